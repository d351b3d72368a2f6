#!/usr/bin/env python3
"""Inverse problem of BASELINE config 5 end to end on one MI355X: 3-D heat equation u_t = kappa (u_xx + u_yy + u_zz) on [0,1]^4 with the
diffusivity kappa estimated (param_estim = true, start 1.0, truth 0.1) from 4096 noisy observations of the analytic solution
exp(-3 pi^2 kappa t) sin(pi x) sin(pi y) sin(pi z).  6x128 tanh MLP; StochasticTraining (interior and boundary sets redrawn on the
device every iteration); the observations enter as a DataLoss term of the same fused evaluation; resident-theta Adam: weights, kappa,
Adam moments, point sets and observations never leave HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads

points = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-3
wl = workloads.cfg5_heat_inverse(points=points, bcs_points=max(4096, points // 16))
rng = np.random.default_rng(7)
kappa_true = 0.1
obs_pts = rng.uniform(size=(4, 4096))
obs = np.exp(-3 * np.pi ** 2 * kappa_true * obs_pts[0]) * np.prod(np.sin(np.pi * obs_pts[1:]), axis=0) + 0.01 * rng.standard_normal(4096)
theta0 = np.concatenate([npde.initialparameters(rng, ch) for ch in wl.chains])
u = wl.pde_system.dvs[0]
disc = npde.PhysicsInformedNN(wl.chains[0], wl.strategy, init_params=theta0, param_estim=True,
                              data_loss=[npde.DataLoss(u, obs_pts, obs, weight=1.0)],
                              adaptive_loss=npde.NonAdaptiveLoss(pde_loss_weights=1.0, bc_loss_weights=10.0, additional_loss_weights=100.0), precision="f32")
prob = npde.discretize(wl.pde_system, disc)
rep = prob.pinnrep
print(rep.engine.describe().split("\n")[0])
theta, done, t_total = rep.flat_init_params, 0, 0.0
print(f"{'iter':>6s} {'loss':>12s} {'kappa':>10s} {'train s':>8s}")
while done < iters:
    n = min(250, iters - done)
    t0 = time.perf_counter()
    res = npde.solve(npde.remake(prob, u0=theta), npde.Adam(lr * 0.5 ** (done // 500)), maxiters=n)        # step halved every 500 iterations
    t_total += time.perf_counter() - t0
    theta, done = res.u, done + n
    print(f"{done:6d} {res.losses[-1]:12.4e} {theta[-1]:10.5f} {t_total:8.2f}", flush=True)
print(f"{iters} iterations on {points} interior + 7 x {max(4096, points // 16)} boundary + 4096 data points: {t_total:.1f} s = "
      f"{t_total / iters * 1e3:.2f} ms/iteration; kappa = {theta[-1]:.4f} (truth {kappa_true})")
