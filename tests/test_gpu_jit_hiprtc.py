"""The compiler-free back end of the runtime specialisation (csrc/jit.cpp, r04): with no hipcc on the machine ($HIPCC points nowhere,
PINN_JIT_BACKEND unset -> hiprtc is chosen automatically) shapes outside the ahead-of-time table are compiled IN PROCESS by libhiprtc from the
headers embedded in the library, cached as code objects (<key>.hsaco + the lowered kernel names), loaded with hipModuleLoadData and launched by
name; the member's SpecInfo comes back from a one-thread kernel of the module.  All three kernel families and a GENERATED jet set, each against
the float64 oracle with the usual 1e-5 bar (tp.check); a second process finds every code object in the cache and compiles nothing."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, os, time
root = %(root)r
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")): sys.path.insert(0, p)
import numpy as np, sympy as sp
import pinn_import
npde = pinn_import.load()
import helpers, pinn_oracle as po, test_emu_parity as tp, test_dgm as td
tp.EXPECTED_BACKEND = "hip"
assert npde._lib.default_library().backend == "hip"
t0 = time.time()
tl = [t0]
def lap(what):
    tl.append(time.time())
    print("  %%-64s %%.1f s" %% (what, tl[-1] - tl[-2]), flush=True)
# family 2 (neuron-split, 64 wide after padding): five hidden layers of 36, mixed second derivatives; GEMM mode switch on the specialised handle
sysm, chain = helpers.shape_problem(npde, 36, 5, 2)
strat = npde.QuasiRandomTraining(40, bcs_points=12, sampling_alg=npde.SobolSample(seed=5), resampling=False, minibatch=1)
rep, prob, sets, th = tp.check(npde, sysm, [chain], strat, tp.theta_for(chain, 61))
assert any("F2_HP64" in l for l in rep.engine.describe().splitlines()), rep.engine.describe()
l0, g0 = rep.engine.loss_grad(th)
rep.engine.set_option("gemm", "fp32")
l1, g1 = rep.engine.loss_grad(th)
assert np.linalg.norm(g1 - g0) < 2e-6 * np.linalg.norm(g0)
lap("5 x 36 net, mixed second derivatives (family 2, both GEMM modes)")
# family 1 (one wave per tile): a 4-input net of width 12, first + pure second derivatives, sigmoid
t, x, y, z = npde.parameters("t x y z")
(u,) = npde.variables("u")
U = u(t, x, y, z)
D = npde.Differential
eq = npde.Eq(D(t)(U), 0.3 * ((D(x) ** 2)(U) + (D(y) ** 2)(U) + (D(z) ** 2)(U)) + U * D(x)(U))
bcs = [npde.Eq(u(0, x, y, z), sp.sin(sp.pi * x) * sp.sin(sp.pi * y) * sp.sin(sp.pi * z)), npde.Eq(u(t, 0, y, z), 0.0)]
dom = [npde.In(v, npde.Interval(0.0, 1.0)) for v in (t, x, y, z)]
chain = npde.Chain(npde.Dense(4, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1))
strat = npde.QuasiRandomTraining(30, bcs_points=12, sampling_alg=npde.SobolSample(seed=6), resampling=False, minibatch=1)
tp.check(npde, npde.PDESystem([eq], bcs, dom, [t, x, y, z], [U]), [chain], strat, tp.theta_for(chain, 62))
lap("3 x 12 sigmoid net of 4 inputs (family 1)")
# a GENERATED jet set (u_xxy, u_xyy: Faa di Bruno rules written by jit.cpp) on a sin net of width 20
x, y = npde.parameters("x y")
U = u(x, y)
Dx, Dy = npde.Differential(x), npde.Differential(y)
eq = npde.Eq(Dx(Dx(Dy(U))) + 0.5 * Dy(Dy(Dx(U))) + U * Dx(Dy(U)) - (Dx ** 2)(U), sp.sin(sp.pi * x) * sp.cos(sp.pi * y))
bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(Dx(u(x, 1)), sp.sin(x))]
dom = [npde.In(v, npde.Interval(0.0, 1.0)) for v in (x, y)]
chain = npde.Chain(npde.Dense(2, 20, "sin"), npde.Dense(20, 20, "sin"), npde.Dense(20, 1))
strat = npde.QuasiRandomTraining(40, bcs_points=10, sampling_alg=npde.SobolSample(seed=7), resampling=False, minibatch=1)
tp.check(npde, npde.PDESystem([eq], bcs, dom, [x, y], [U]), [chain], strat, tp.theta_for(chain, 63), mode="exact")
lap("2 x 20 sin net, generated jet set u_xxy, u_xyy (family 1)")
# family 3 (DGM): 22 modes, 2 gated layers
net = npde.DGM(2, 1, 22, 2, "tanh", "tanh", "identity")
strat = npde.QuasiRandomTraining(70, bcs_points=20, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1)
rep, prob, sets, th = tp.check(npde, td._burgers(npde), [net], strat, tp.theta_for(net, 64), weights=[1.0, 2.0, 0.5, 3.0], mode="exact")
assert all("F3_" in l for l in rep.engine.describe().splitlines() if "kernel=" in l)
lap("DGM 22 modes x 2 gated layers (family 3)")
# the shape the round-3 review asked a cold create of: a 3 x 200 tanh chain (padded to 256, fp32 MFMA kernels): pinn_create alone, then the check
sysm, _ = tp.poisson2d(npde)
chain = npde.Chain(npde.Dense(2, 200, "tanh"), npde.Dense(200, 200, "tanh"), npde.Dense(200, 200, "tanh"), npde.Dense(200, 1))
mk = lambda: npde.QuasiRandomTraining(24, bcs_points=70, sampling_alg=npde.SobolSample(seed=9), resampling=False, minibatch=1)     # (> 64 boundary points: their own launch)
tc = time.time()
prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, mk(), init_params=tp.theta_for(chain, 65), precision="f32"))
print("  COLD_CREATE 3 x 200 chain (two members: interior forward-Laplacian set + value-only boundary set): %%.2f s" %% (time.time() - tc), flush=True)
tl[-1] = time.time()
tp.check(npde, sysm, [chain], mk(), tp.theta_for(chain, 65))
lap("3 x 200 tanh chain: second handle + evaluation + oracle check")
print("HIPRTC_OK %%.1f s" %% (time.time() - t0))
'''


def _run(env):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=1500)


def test_hiprtc_backend_without_a_compiler(npde, hip_lib, tmp_path):
    cache = tmp_path / "jit"
    cache.mkdir(mode=0o700)
    env = dict(os.environ, PINN_JIT_DIR=str(cache), HIPCC=str(tmp_path / "no_such_hipcc"), PINN_SRC_DIR=str(tmp_path / "nowhere"))
    env.pop("PINN_JIT_BACKEND", None)
    r = _run(env)
    assert "HIPRTC_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
    assert "with hiprtc" in r.stderr                                               # (compiled in process ...)
    files = [f for _, _, fs in os.walk(cache) for f in fs]
    hsaco = [f for f in files if f.endswith(".hsaco")]
    assert len(hsaco) >= 6 and not [f for f in files if f.endswith(".so")], files          # (... into code objects, no shared object, nothing unpacked)
    assert not any(f.endswith(".hpp") for f in files), files
    print("first process:\n" + "\n".join(l for l in r.stdout.splitlines() if l.startswith("  ") or "HIPRTC_OK" in l), "\n ", len(hsaco), "code objects")
    r2 = _run(env)
    assert "HIPRTC_OK" in r2.stdout, r2.stdout[-3000:] + r2.stderr[-6000:]
    assert "specialising" not in r2.stderr, r2.stderr[-2000:]                       # every member came out of the cache
    print("second process:", r2.stdout.strip().splitlines()[-1])
