#!/usr/bin/env python3
"""Single-GPU timing of all five BASELINE.json configurations at their full sizes (parity for them: tests/test_gpu_parity.py).
Reports ms per loss+gradient evaluation, interior-point evals/s and EXECUTED TFLOP/s (SURVEY.md §8d flop model applied to the jet channels the kernels actually carry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
from bench import algorithmic_flops_per_point, PEAK_FP32_MFMA_TFLOPS

if os.environ.get("PINN_AB_LIB"):            # A/B runs: time another build of the library (tools/ab_compare.py convention)
    npde._lib.set_library(npde.Library(os.environ["PINN_AB_LIB"]))
scale = float(os.environ.get("SCALE", "1.0"))
cfgs = [("cfg1", workloads.cfg1_poisson1d, dict(points=1024)),
        ("cfg2", workloads.cfg2_poisson2d, dict(points=65536)),
        ("cfg3", workloads.cfg3_burgers, dict(points=int(262144 * scale))),
        ("cfg4", workloads.cfg4_cavity, dict(points=int(262144 * scale), bcs_points=int(32768 * scale))),
        ("cfg5", workloads.cfg5_heat_inverse, dict(points=int(1000000 * scale), bcs_points=int(65536 * scale)))]
only = sys.argv[1:] or [c[0] for c in cfgs]
print(f"{'config':44s} {'P':>7s} {'points':>9s} {'ms/eval':>9s} {'int-pts/s':>11s} {'TFLOP/s':>8s} {'%peak':>6s}   kernels (ms)")
for name, fn, kw in cfgs:
    if name not in only:
        continue
    wl = fn(**kw)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    P, K = eng.P, eng.K
    th = torch.tensor(rep.flat_init_params, dtype=torch.float32, device="cuda")
    out = torch.zeros(P + K, dtype=torch.float32, device="cuda")
    w = None
    if wl.adaptive_loss is not None:
        al = wl.adaptive_loss
        al.broadcast(len(rep.pde_train_sets), len(rep.bcs_train_sets))
        w = list(al.pde_loss_weights) + list(al.bc_loss_weights)
    st = torch.cuda.current_stream()
    eng.set_timing(1, -1)                      # HIP events around every launch group for the per-kernel column
    for _ in range(3):
        eng.loss_grad_device(th.data_ptr(), out.data_ptr(), w, st.cuda_stream)
    torch.cuda.synchronize()
    groups = eng.group_timings()
    eng.set_timing(0, -1)
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        eng.loss_grad_device(th.data_ptr(), out.data_ptr(), w, st.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    # algorithmic flops: every launch group = one network's jet set over that group's points
    flops = 0.0
    for g in groups:
        net = 0
        sizes = wl.chains[min(net, len(wl.chains) - 1)].sizes
        flops += algorithmic_flops_per_point(sizes, g["channels"]) * g["points"]
    n_int = sets[0].shape[1]
    tf = flops / (ms * 1e-3) / 1e12
    print(f"{wl.name:44s} {P:7d} {sum(s.shape[1] for s in sets):9d} {ms:9.3f} {n_int / (ms * 1e-3):11.3e} {tf:8.1f} {100 * tf / PEAK_FP32_MFMA_TFLOPS:6.1f}   "
          + " ".join(f"C{g['channels']}:{g['ms']:.3f}" for g in groups), flush=True)
    print("    " + eng.describe().replace("\n", "\n    ").rstrip(), flush=True)
    del rep, eng
