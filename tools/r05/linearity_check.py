"""r05: gradient linearity in the term weights at full size (tests/test_gpu_parity.py::test_full_size_cfg2_properties (b)) — both sides against the float64 mode"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
w = np.array([1.0, 2.0, 0.5, 4.0, 3.0], dtype=np.float32)
wl = workloads.cfg2_poisson2d(points=65536)
rep = m.symbolic_discretize(wl.pde_system, wl.discretization())         # the float64 reference: the product library's float64 mode
eng = rep.engine
eng.set_option("precision", "f64")
th = np.asarray(wl.theta, dtype=np.float32).astype(np.float64)
L64, G64 = eng.loss_grad_f64(th, w.astype(np.float64))
tg64 = []
for k in range(eng.K):
    e = np.zeros(eng.K); e[k] = 1.0
    tg64.append(eng.loss_grad_f64(th, e)[1])
tg64 = np.array(tg64)
del rep, eng
for name in sys.argv[1:] or ["head"]:
    m._lib.set_library(None if name == "head" else m.Library(os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", f"libpinn_{name}.so")))
    wl = workloads.cfg2_poisson2d(points=65536)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    L, G = eng.loss_grad(wl.theta, w)
    Lt, TG = eng.term_grads(wl.theta)
    Gw = (w[:, None].astype(np.float64) * TG.astype(np.float64)).sum(axis=0)
    n = np.linalg.norm
    print(f"{name}: |G| = {n(G64):.3f}; per-term |TG_k| = " + " ".join(f"{n(t):.1f}" for t in tg64))
    print(f"   weighted evaluation vs float64: {n(G - G64) / n(G64):.2e};  sum_k w_k TG_k vs float64: {n(Gw - G64) / n(G64):.2e};  the two against each other: {n(Gw - G) / n(G):.2e}")
    print("   per-term gradients vs float64 (relative to the term's own norm): " + " ".join(f"{n(TG[k] - tg64[k]) / n(tg64[k]):.2e}" for k in range(eng.K)))
    del rep, eng
