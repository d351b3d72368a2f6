"""Times the FLOAT64 evaluation mode (pinn_set_option(h, "precision", "f64")) against the fp32 kernels on the same
handle: the bench workload (2-D Poisson 4x64, 65,536 + 4x65,536 points), the reference's own regime (cfg1: 3x32... small) and the 4-D inverse
heat problem reduced to 65,536 points.  Host-entry wall time per fused loss + gradient (theta over PCIe both ways, as a quasi-Newton caller
pays it) and the distance between the two evaluations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def timed(fn, n):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    import pinn_import
    npde = pinn_import.load()
    from neuralpde_jl_amd import workloads
    cases = [("cfg1 1-D Poisson, 3x32... as in BASELINE.json, 4,096 points", lambda: workloads.cfg1_poisson1d(4096)),
             ("cfg2 2-D Poisson 4x64, 65,536 + 4x65,536 points (bench workload)", lambda: workloads.cfg2_poisson2d(points=65536)),
             ("cfg3 Burgers 4x64, 65,536 + 3x8,192 points", lambda: workloads.cfg3_burgers(points=65536, bcs_points=8192)),
             ("cfg4 cavity 3 x (5x128), 16,384 + 8x4,096 points", lambda: workloads.cfg4_cavity(points=16384, bcs_points=4096)),
             ("cfg5 inverse heat 6x128 d=4, 32,768 + 7x8,192 points", lambda: workloads.cfg5_heat_inverse(points=32768, bcs_points=8192))]
    only = sys.argv[1:]                                           # (case prefixes, e.g. `cfg3`: that case alone in a fresh process)
    if only:
        cases = [c for c in cases if any(c[0].startswith(o) for o in only)]
    for name, make in cases:
        wl = make()
        rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
        eng = rep.engine
        assert eng.L.backend == "hip"
        th = np.asarray(rep.flat_init_params, dtype=np.float64)
        npts = sum(s.shape[1] for s in rep.pde_train_sets + rep.bcs_train_sets)
        l32, g32 = eng.loss_grad_f64(th)
        t32 = timed(lambda: eng.loss_grad_f64(th), 20)
        eng.set_option("precision", "f64")
        l64, g64 = eng.loss_grad_f64(th)
        t64 = timed(lambda: eng.loss_grad_f64(th), 5)
        path = eng.get_option("f64_path")
        tl = None
        if path != "lanes":                                  # r05: the same evaluation on the one-lane-per-point kernels (family 4) for comparison
            os.environ["PINN_F64_NO_MFMA"] = "1"
            ll, gl = eng.loss_grad_f64(th)
            tl = timed(lambda: eng.loss_grad_f64(th), 3)
            del os.environ["PINN_F64_NO_MFMA"]
            print(f"   [{name.split()[0]}] float64 kernels: {path} {t64:.3f} ms vs one lane per point {tl:.3f} ms ({tl / t64:.1f}x); the two agree to "
                  f"{np.linalg.norm(gl - g64) / np.linalg.norm(g64):.1e} (gradient rel L2)", flush=True)
        print(f"{name}: P = {eng.P}, {npts} points\n   fp32 kernels {t32:8.3f} ms  ({npts / t32 * 1e3:.3e} evals/s)    float64 mode {t64:8.3f} ms  ({npts / t64 * 1e3:.3e} evals/s)   ratio {t64 / t32:.1f}x"
              f"\n   fp32 vs float64: loss rel {np.max(np.abs(l32 - l64) / np.abs(l64)):.2e}, gradient rel L2 {np.linalg.norm(g32 - g64) / np.linalg.norm(g64):.2e}", flush=True)


if __name__ == "__main__":
    main()
