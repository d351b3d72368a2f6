#!/usr/bin/env python3
"""Per-phase cycle breakdown of the family-2 kernels (profiling build from tools/stamp_profile.sh; GPU box only)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
PH = ["0 coords+layer0+act0", "1 publish+barrier (fwd)", "2 fwd GEMM (+W/b loads)", "3 act_forward (+record st)", "4 out layer+barrier+U",
      "5 tape", "6 out adjoint+act_adj(last)", "7 record ld+publish+stage", "8 barrier (staged)", "9 dW GEMM", "10 dA GEMM",
      "11 barrier (X free)", "12 act_adjoint", "13 layer-0 grads", "14 epilogue", "15 loop top", "16 tape: setup (zero, inputs, sources)", "17 tape: forward ops", "18 tape: adjoint ops", "19 tape: seeds -> LDS", "20", "21", "22", "23"]
tag = sys.argv[1] if len(sys.argv) > 1 else ""
path = os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", f"libpinn_stamp{tag}.so")
lib = m.Library(path)
m._lib.set_library(lib)
pts = int(os.environ.get("POINTS", "65536"))
wl = workloads.cfg2_poisson2d(points=pts)
rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
eng = rep.engine
eng.set_timing(1, -1)
for _ in range(5):
    eng.loss_grad(wl.theta)
gt = eng.group_timings()
dbg = lib.lib.pinn_debug_slab
dbg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int64]
for g, info in enumerate(gt):
    nb = dbg(eng.h, g, -1, None, 0)
    acc = np.zeros((4, 24))
    buf = np.zeros(128, dtype=np.float32)
    nsample = min(nb, 64)
    for b in np.linspace(0, nb - 1, nsample).astype(int):
        assert dbg(eng.h, g, int(b), buf.ctypes.data_as(C.POINTER(C.c_float)), 128) == 0
        acc += buf.view(np.uint32)[:96].reshape(4, 24)
    acc /= nsample
    tiles_per_wg = info["tiles"] / nb
    tot = acc.sum(axis=1)
    print(f"group {g}: C={info['channels']} tiles={info['tiles']} blocks={nb} kernel {info['ms']*1e3:.1f} us; per-WG total cycles (wave 0..3): "
          + " ".join(f"{t:.0f}" for t in tot) + f"  = {tot[0]/info['ms']/1e3:.0f} MHz counter")
    print(f"  {'phase':32s} " + " ".join(f"{'w'+str(w):>9s}" for w in range(4)) + "   share(w0)   cyc/tile(w0)")
    for i, name in enumerate(PH):
        if not acc[:, i].any():
            continue
        print(f"  {name:32s} " + " ".join(f"{acc[w, i]:9.0f}" for w in range(4)) + f"   {100*acc[0,i]/tot[0]:6.1f} %   {acc[0,i]/tiles_per_wg:9.0f}")
