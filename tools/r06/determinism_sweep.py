"""r06 (second half): bit-reproducibility of the PRODUCT kernels, stand-alone AND merged launches, `reps` repetitions each:
every BASELINE configuration (reduced and full size), the shape grid the parity tests use (widths 16 .. 128, 1 .. 5 hidden layers, 1 .. 3 inputs:
the ahead-of-time instantiations of families 1 and 2 plus the run-time specialised shapes), both GEMM arithmetics of family 2, and the float64 mode
(families 4m / 4s).  Per case: loss + gradient of the whole problem (merged / chained launches) and every term on its own (pinn_term_grads: one
stand-alone launch per term) against the first repetition, bitwise.   python tools/r06/determinism_sweep.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
import helpers
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def shape_case(width, hidden, d):
    sysm, chain = helpers.shape_problem(m, width, hidden, d)
    strat = m.QuasiRandomTraining(4096, bcs_points=512, sampling_alg=m.SobolSample(seed=3), resampling=False, minibatch=1)
    return workloads.Workload(f"shape {width}x{hidden} d{d}", sysm, [chain], strat, workloads.synthetic_theta([chain], 100 + width + hidden + d))


cases = [("cfg1 3x32 1,024", lambda: workloads.cfg1_poisson1d(1024)),
         ("cfg2 4x64 8,192 + 4x8,192", lambda: workloads.cfg2_poisson2d(points=8192)),
         ("cfg2 4x64 65,536 + 4x65,536 (bench)", lambda: workloads.cfg2_poisson2d(points=65536)),
         ("cfg3 4x64 262,144 + 3x262,144", lambda: workloads.cfg3_burgers(points=262144)),
         ("cfg4 3x(5x128) 16,384 + 8x4,096", lambda: workloads.cfg4_cavity(points=16384, bcs_points=4096)),
         ("cfg5 6x128 d4 32,768 + 7x8,192", lambda: workloads.cfg5_heat_inverse(points=32768, bcs_points=8192))]
for width in (16, 32, 64, 128):
    for hidden, d in ((1, 1), (2, 2), (3, 3), (4, 2), (5, 1)):
        if width >= 64 and hidden < 2:
            continue                                                   # (the neuron-split kernels need two hidden layers)
        cases.append((f"shape {width} x {hidden}, d = {d}", lambda w=width, h=hidden, dd=d: shape_case(w, h, dd)))


def run(eng, th, f64):
    ev = (lambda: eng.loss_grad_f64(th)) if f64 else (lambda: eng.loss_grad(th))
    tg = (lambda: eng.term_grads_f64(th)) if f64 else (lambda: eng.term_grads(th))
    l0, g0 = ev()
    L0, T0 = tg()
    bad_m = bad_t = 0
    for _ in range(reps):
        l, g = ev()
        bad_m += int(not (np.array_equal(l, l0) and np.array_equal(g, g0)))
    for _ in range(max(reps // 4, 5)):                                 # (K launches per repetition)
        L, T = tg()
        bad_t += int(not (np.array_equal(L, L0) and np.array_equal(T, T0)))
    return bad_m, bad_t


print(f"{'case':44s} {'mode':6s} merged evaluation / stand-alone per-term launches differing from the first (of {reps} / {max(reps // 4, 5)})")
tot = 0
for name, mk in cases:
    wl = mk()
    t0 = time.time()
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    th = np.asarray(rep.flat_init_params, dtype=np.float32)
    row = []
    for mode in ("split", "fp32"):
        eng.set_option("gemm", mode)
        bm, bt = run(eng, th, False)
        row.append((mode, bm, bt)); tot += bm + bt
    try:
        eng.set_option("gemm", "split")
        eng.set_option("precision", "f64")
        bm, bt = run(eng, th.astype(np.float64), True)
        row.append(("f64:" + eng.get_option("f64_path"), bm, bt)); tot += bm + bt
    except m.EngineError as e:
        row.append(("f64", "n/a", str(e)[:40]))
    print(f"{name:44s} " + "   ".join(f"{md}: {a} / {b}" for md, a, b in row) + f"   ({time.time() - t0:.0f} s)", flush=True)
    del rep, eng
print("ALL BIT-REPRODUCIBLE" if tot == 0 else f"NON-REPRODUCIBLE repetitions: {tot}")
