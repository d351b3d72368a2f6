"""r05 item 1: decomposition of the trained-theta error of the fp32 engine (CPU: emulation library; GPU: product library).
usage: python tools/r05/theta_ab.py [--lib path] [--points N] [--tags adam2000,adam6000] [--cfg cfg2|cfg3]
Prints engine error vs exact float64 oracle and torch-f32 error vs the same, on a reduced point set (the oracle is evaluated here)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pinn_import
m = pinn_import.load()
import pinn_oracle as po
import helpers
from neuralpde_jl_amd import workloads, symbolic
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--points", type=int, default=4096)
ap.add_argument("--bcs", type=int, default=1024)
ap.add_argument("--tags", default="adam2000,adam6000")
ap.add_argument("--cfg", default="cfg2")
ap.add_argument("--modes", default="split,fp32")
ap.add_argument("--round-theta", action="store_true", help="reference = float64 oracle at the float32-rounded parameters (what an fp32 engine is handed)")
a = ap.parse_args()
if a.lib:
    m._lib.set_library(m.Library(a.lib))
g = np.load(os.path.join(ROOT, "tests", "golden", a.cfg + "_variants.npz"))
wl = workloads.cfg2_poisson2d(points=a.points, bcs_points=a.bcs) if a.cfg == "cfg2" else workloads.cfg3_burgers(points=a.points, bcs_points=a.bcs)
rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
eng = rep.engine
sets = rep.pde_train_sets + rep.bcs_train_sets
prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
w = g["weights"]
def err(l, gr, lr, gref):
    return np.max(np.abs(l - lr) / np.abs(lr)), np.linalg.norm(gr - gref) / np.linalg.norm(gref), np.max(np.abs(gr - gref)) / np.max(np.abs(gref))
for tag in a.tags.split(","):
    th = g["theta_" + tag]
    ex = po.loss_and_grad(prob, th, sets, weights=w, mode="exact")
    if a.round_theta:
        ex0 = ex
        ex = po.loss_and_grad(prob, th.astype(np.float32).astype(np.float64), sets, weights=w, mode="exact")
        e0 = err(ex0.term_losses, ex0.grad, ex.term_losses, ex.grad)
        print(f"{a.cfg} {tag}: oracle(theta64) vs oracle(float32(theta64)) — the input quantisation alone: loss {e0[0]:.2e} grad L2 {e0[1]:.2e} Linf {e0[2]:.2e}")
    po.DT = torch.float32
    f32 = po.loss_and_grad(prob, th, sets, weights=w, mode="exact")
    po.DT = torch.float64
    ef = err(f32.term_losses, f32.grad, ex.term_losses, ex.grad)
    print(f"{a.cfg} {tag} N={a.points}+{a.bcs}: |grad|={np.linalg.norm(ex.grad):.3e}  torch-f32: loss {ef[0]:.2e} grad L2 {ef[1]:.2e} Linf {ef[2]:.2e}", flush=True)
    for mode in a.modes.split(","):
        eng.set_option("gemm", mode)
        t0 = time.time()
        l, gr = eng.loss_grad(th, w)
        e = err(l, gr, ex.term_losses, ex.grad)
        print(f"   engine gemm={mode:5s}: loss {e[0]:.2e} grad L2 {e[1]:.2e} Linf {e[2]:.2e}   ratio to torch-f32: {e[0]/ef[0]:.2f} / {e[1]/ef[1]:.2f} / {e[2]/ef[2]:.2f}   ({time.time()-t0:.1f}s)", flush=True)
