"""Times pinn_term_grads (K per-term gradients) against pinn_loss_grad on the bench workload (2-D Poisson 4x64, 65,536 + 4x65,536 points)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

def main():
    import pinn_import
    npde = pinn_import.load()
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=65536)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    assert eng.L.backend == "hip"
    theta = np.asarray(wl.theta, dtype=np.float32)
    l0, g0 = eng.loss_grad(theta)
    l1, tg = eng.term_grads(theta)
    print("sum of per-term gradients vs gradient: rel", float(np.abs(tg.sum(0) - g0).max() / np.abs(g0).max()), " losses rel",
          float(np.abs(l1 - l0).max() / np.abs(l0).max()))
    for name, fn in (("loss_grad", lambda: eng.loss_grad(theta)), ("term_grads", lambda: eng.term_grads(theta))):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(100): fn()
        print(f"{name}: {(time.perf_counter() - t0) * 10:.3f} ms / call (K = {eng.K})")

if __name__ == "__main__":
    main()
