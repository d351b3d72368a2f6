import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import pinn_import; m = pinn_import.load()
import pinn_oracle as po, helpers
from neuralpde_jl_amd import workloads
wl = workloads.cfg2_poisson2d(points=16, bcs_points=64)
rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
sets = rep.pde_train_sets + rep.bcs_train_sets
prob = helpers.oracle_problem(m, wl.pde_system, wl.chains)
reft = po.loss_and_grad(prob, wl.theta, sets, mode='stencil', per_term_grads=True)
sizes = wl.chains[0].sizes
for trial in range(3):
    lt, tg = rep.engine.term_grads(wl.theta)
    g, r = tg[0].astype(np.float64), reft.term_grads[0]
    off = 0
    out = []
    for j in range(len(sizes)-1):
        nin, nout = sizes[j], sizes[j+1]
        W = slice(off, off+nin*nout); off += nin*nout
        b = slice(off, off+nout); off += nout
        out.append((j, float(np.abs(g[W]-r[W]).max()/np.abs(r[W]).max()), float(np.abs(g[b]-r[b]).max()/max(np.abs(r[b]).max(),1e-30))))
    print('trial', trial, ' '.join(f"L{j}: W {ew:.1e} b {eb:.1e}" for j,ew,eb in out))
    if trial == 0:
        j=2; nin,nout=64,64; o=sum(sizes[i]*sizes[i+1]+sizes[i+1] for i in range(j))
        Wg=g[o:o+4096].reshape(nin,nout).T; Wr=r[o:o+4096].reshape(nin,nout).T
        E=np.abs(Wg-Wr)/np.abs(Wr).max()
        print('  layer2 W err by out-row block', [float(E[16*k:16*k+16].max()) for k in range(4)], 'by in-col block', [float(E[:,16*k:16*k+16].max()) for k in range(4)])
