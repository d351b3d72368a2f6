"""r06: the hardware-only non-determinism of the REMOVED ti-outer dW variant (PINN_F2_SPLIT_ACC2 bit 16, commit 91b4857
removed it) — rebuilt in a scratch tree from the removal diff, once as it was and once each with  -mllvm -amdgpu-waitcnt-forcezero,
-DPINN_STORE_PAD=16  (store_pad widened) and  -fno-slp-vectorize.  Per library: `reps` stand-alone evaluations of every term's gradient
(pinn_term_grads: one launch per term) and of the merged evaluation; how many repetitions differ from the first one bitwise, and by how much.
Usage: python tools/r06/ti_outer_determinism.py [reps] lib1 lib2 ...   (names of neuralpde.jl_amd/csrc/abl/libpinn_<name>.so; `head` = the product)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
args = sys.argv[1:]
reps = int(args[0]) if args and args[0].isdigit() else 40
names = [a for a in args if not a.isdigit()] or ["head"]
for name in names:
    m._lib.set_library(None if name == "head" else m.Library(os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", f"libpinn_{name}.so")))
    wl = workloads.cfg2_poisson2d(points=65536)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    L0, T0 = eng.term_grads(wl.theta)
    l0, g0 = eng.loss_grad(wl.theta)
    bad_t, bad_m, worst = np.zeros(eng.K, dtype=int), 0, np.zeros(eng.K)
    for i in range(reps):
        L, T = eng.term_grads(wl.theta)
        for k in range(eng.K):
            if not np.array_equal(T[k], T0[k]):
                bad_t[k] += 1
                worst[k] = max(worst[k], np.linalg.norm(T[k].astype(np.float64) - T0[k]) / np.linalg.norm(T0[k].astype(np.float64)))
        l, g = eng.loss_grad(wl.theta)
        bad_m += int(not (np.array_equal(g, g0) and np.array_equal(l, l0)))
    print(f"{name:>10s}: stand-alone per-term launches differing from the first of {reps}: {bad_t.tolist()} (worst rel L2 {['%.1e' % x for x in worst]}); merged evaluation: {bad_m} of {reps}", flush=True)
    del rep, eng
