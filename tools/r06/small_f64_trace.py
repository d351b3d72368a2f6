"""a small problem (poisson2d 2x16 on 165 points, K = 5) evaluated 40 times in float64 mode + 40 resident Adam iterations: the command behind the
rocprofv3 kernel trace of profiles/r06_small_f64.txt.  usage: tools/trace_cmd.sh <out> /abs/path/tools/r06/small_f64_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.chdir(ROOT)
import numpy as np
import pinn_import
npde = pinn_import.load()
import test_emu_parity as tp
sysm, chain = tp.poisson2d(npde, "tanh", width=16, hidden=2)
rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.1), init_params=tp.theta_for(chain, 5)))
eng = rep.engine
print(eng.describe())
th = np.asarray(rep.flat_init_params, dtype=np.float64)
for _ in range(40):
    eng.loss_grad_f64(th)
eng.adam_f64(th, 40, 1e-3)
