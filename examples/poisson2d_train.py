#!/usr/bin/env python3
"""End-to-end training of the BASELINE workload on one MI355X: 2-D Poisson (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:65-89),
4x64 tanh MLP, 65,536 interior + 4 x 65,536 boundary points, resident-theta Adam (`solve(prob, Adam(lr); maxiters)` mirror:
theta, moments and point sets never leave HBM).  Prints the loss history and the error against the analytic solution
u(x, y) = sin(pi x) sin(pi y) / (2 pi^2) on a 100 x 100 grid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-3
decay = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6          # step-size factor per 1000 iterations
wl = workloads.cfg2_poisson2d(points=65536)
rng = np.random.default_rng(0)
theta0 = np.concatenate([npde.initialparameters(rng, ch) for ch in wl.chains])          # glorot weights, zero biases (Lux default)
disc = npde.PhysicsInformedNN(wl.chains[0], wl.strategy, init_params=theta0,
                              adaptive_loss=npde.NonAdaptiveLoss(pde_loss_weights=1.0, bc_loss_weights=100.0), precision="f32")
prob = npde.discretize(wl.pde_system, disc)
xs = np.linspace(0.0, 1.0, 100)
X, Y = np.meshgrid(xs, xs, indexing="ij")
grid = np.stack([X.ravel(), Y.ravel()])
exact = np.sin(np.pi * grid[0]) * np.sin(np.pi * grid[1]) / (2 * np.pi ** 2)
theta, done, t_total = theta0, 0, 0.0
print(f"{'iter':>6s} {'loss':>12s} {'max|u - u*|':>12s} {'train s':>8s}")
while done < iters:
    n = min(1000, iters - done)
    p = npde.remake(prob, u0=theta)
    t0 = time.perf_counter()
    res = npde.solve(p, npde.Adam(lr * (decay ** (done // 1000))), maxiters=n)      # (every solve starts Adam afresh, as solve(remake(prob, u0 = res.u)) does in the reference)
    t_total += time.perf_counter() - t0
    theta, done = res.u, done + n
    u = prob.pinnrep.phi(grid, theta)[0]
    print(f"{done:6d} {res.losses[-1]:12.4e} {np.max(np.abs(u - exact)):12.4e} {t_total:8.2f}", flush=True)
print(f"{iters} Adam iterations on 65,536 + 4 x 65,536 points: {t_total:.2f} s = {t_total / iters * 1e3:.3f} ms/iteration "
      f"({65536 * iters / t_total:.3e} interior-point evals/s incl. the optimiser step)")
