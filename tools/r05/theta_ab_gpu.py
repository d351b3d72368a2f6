"""r05 item 1: the trained-parameter error of the fp32 engine at FULL size on the GPU, per library variant and GEMM mode, against the committed
float64 fixtures (tests/golden/cfg{2,3}_variants.npz: exact-derivative oracle at the float64 parameters + torch-float32 evaluation of the same program).
usage: python tools/r05/theta_ab_gpu.py name1 name2 ...      (names of neuralpde.jl_amd/csrc/abl/libpinn_<name>.so; `head` = the product library)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
names = sys.argv[1:] or ["head"]
CASES = [("cfg2", lambda: workloads.cfg2_poisson2d(points=65536), ("adam2000", "adam6000", "x2")),
         ("cfg3", lambda: workloads.cfg3_burgers(points=262144), ("adam2000",))]
def err(l, gr, lr, gref):
    return np.array([np.max(np.abs(l - lr) / np.abs(lr)), np.linalg.norm(gr - gref) / np.linalg.norm(gref), np.max(np.abs(gr - gref)) / np.max(np.abs(gref))])
for cfg, make, tags in CASES:
    g = np.load(os.path.join(ROOT, "tests", "golden", cfg + "_variants.npz"))
    w = g["weights"]
    for tag in tags:
        lr, gref = g[f"losses_exact_{tag}"], g[f"grad_exact_{tag}"]
        ef = err(g[f"losses_f32_{tag}"], g[f"grad_f32_{tag}"], lr, gref)
        has32 = f"grad_exact32_{tag}" in g
        print(f"{cfg} {tag} (full size): |grad| = {np.linalg.norm(gref):.3e}; torch-f32 vs exact float64 oracle: loss {ef[0]:.2e} grad L2 {ef[1]:.2e} Linf {ef[2]:.2e}", flush=True)
        if has32:
            ef2 = err(g[f"losses_f32_{tag}"], g[f"grad_f32_{tag}"], g[f"losses_exact32_{tag}"], g[f"grad_exact32_{tag}"])
            print(f"    torch-f32 vs oracle at float32(theta): loss {ef2[0]:.2e} grad L2 {ef2[1]:.2e} Linf {ef2[2]:.2e}  per-term loss errors " +
                  " ".join(f"{x:.1e}" for x in np.abs(g[f"losses_f32_{tag}"] - g[f"losses_exact32_{tag}"]) / np.abs(g[f"losses_exact32_{tag}"])) +
                  "   term losses " + " ".join(f"{x:.2e}" for x in g[f"losses_exact32_{tag}"]))
            e32 = err(lr, gref, g[f"losses_exact32_{tag}"], g[f"grad_exact32_{tag}"])
            print(f"    input quantisation alone (oracle at theta64 vs oracle at float32(theta64)): loss {e32[0]:.2e} grad L2 {e32[1]:.2e} Linf {e32[2]:.2e}")
        for name in names:
            m._lib.set_library(None if name == "head" else m.Library(os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", f"libpinn_{name}.so")))
            wl = make()
            rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
            eng = rep.engine
            for mode in ("split", "fp32"):
                eng.set_option("gemm", mode)
                l, gr = eng.loss_grad(g["theta_" + tag], w)
                e = err(l, gr, lr, gref)
                line = f"    {name:>8s} gemm={mode:5s}: loss {e[0]:.2e} grad L2 {e[1]:.2e} Linf {e[2]:.2e}   x torch-f32: {e[0]/ef[0]:.2f} / {e[1]/ef[1]:.2f} / {e[2]/ef[2]:.2f}"
                if has32:
                    e2 = err(l, gr, g[f"losses_exact32_{tag}"], g[f"grad_exact32_{tag}"])
                    line += f"   | vs oracle at float32(theta): loss {e2[0]:.2e} grad L2 {e2[1]:.2e} Linf {e2[2]:.2e}"
                    line += "  per-term loss errors " + " ".join(f"{x:.1e}" for x in np.abs(l - g[f"losses_exact32_{tag}"]) / np.abs(g[f"losses_exact32_{tag}"]))
                print(line, flush=True)
            del rep, eng
