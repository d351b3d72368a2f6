"""Throughput of the small-net kernels (family 1: one wave per tile) on a large point set: 2-D Poisson, width x hidden nets, 262,144 interior +
4 x 65,536 boundary points."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
from bench import algorithmic_flops_per_point, PEAK_FP32_MFMA_TFLOPS
for width, hidden in ((16, 2), (16, 3), (32, 2), (32, 3), (64, 3)):
    wl = workloads.cfg2_poisson2d(points=262144, bcs_points=65536, width=width, hidden=hidden)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    th = torch.tensor(rep.flat_init_params, dtype=torch.float32, device="cuda"); out = torch.zeros(eng.P + eng.K, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(20): eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    sizes = wl.chains[0].sizes
    fl = 262144 * algorithmic_flops_per_point(sizes, 4) + 4 * 65536 * algorithmic_flops_per_point(sizes, 1)
    k = [l.split("kernel=")[1].split()[0] for l in eng.describe().splitlines() if "kernel=" in l]
    print(f"{width:3d} x {hidden}: {dt * 1e3:7.3f} ms/eval  {262144 / dt:.3e} interior pts/s  {fl / dt / 1e12:6.1f} TFLOP/s executed = {fl / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS * 100:4.1f} % of fp32 MFMA peak   {k}")
