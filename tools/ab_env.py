"""A/B of the launch structure of one evaluation inside ONE process on one GPU box (the engine reads these switches at every call):
    merged     : default — interior + boundary tiles in one persistent launch, one-kernel reduction
    chained    : PINN_NO_MERGE=1 — two chained launches (round 2), one-kernel reduction
    chained+2st: PINN_NO_MERGE=1 PINN_NO_REDUCE_ONE=1 — two chained launches, reduce1 + reduce2 (the round-2 launch sequence)
    merged+2st : PINN_NO_REDUCE_ONE=1
    loss-only  : pinn_loss_device (MODE_LOSS launches, K sums only)
Usage: python tools/ab_env.py [--cfg cfg2] [--points N ...] [--steps 300]; prints median / min wall time per evaluation (result delivered to
the host every step as bench.py does), no HIP events."""
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg2")
ap.add_argument("--points", type=int, nargs="*", default=[65536, 8192])
ap.add_argument("--steps", type=int, default=300)
args = ap.parse_args()
VARIANTS = [("merged", {}), ("chained", {"PINN_NO_MERGE": "1"}),
            ("chained+2st", {"PINN_NO_MERGE": "1", "PINN_NO_REDUCE_ONE": "1"}), ("merged+2st", {"PINN_NO_REDUCE_ONE": "1"})]
if os.environ.get("PINN_LIB"):                      # A/B of another build of the library (neuralpde.jl_amd/csrc/abl/libpinn_<name>.so)
    m._lib.set_library(m.Library(os.environ["PINN_LIB"]))
for pts in args.points:
    wl = workloads.CONFIGS[args.cfg](points=pts)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    th = torch.tensor(np.asarray(rep.flat_init_params, dtype=np.float32), device="cuda")
    out_h = torch.zeros(eng.P + eng.K, dtype=torch.float32).pin_memory()
    sums_h = torch.zeros(eng.K, dtype=torch.float32).pin_memory()
    st = torch.cuda.current_stream()
    eng.set_timing(0, -1)
    res = {}
    for rnd in range(3):
        for tag, env in VARIANTS + [("loss-only", {})]:
            for k in ("PINN_NO_MERGE", "PINN_NO_REDUCE_ONE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            ts = []
            for i in range(args.steps + 30):
                t0 = time.perf_counter()
                if tag == "loss-only":
                    eng.loss_device(th.data_ptr(), sums_h.data_ptr(), st.cuda_stream)
                else:
                    eng.loss_grad_device(th.data_ptr(), out_h.data_ptr(), None, st.cuda_stream)
                st.synchronize()
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts[30:]) * 1e6
            res.setdefault(tag, []).append((np.median(ts), ts.min(), float(out_h[:eng.P].double().sum()), out_h[eng.P:].numpy().copy(), sums_h.numpy().copy()))
    print(f"== {wl.name} points={pts} terms={eng.K}: {eng.describe().splitlines()[1][:110]}")
    for tag, r in res.items():
        print(f"  {tag:12s} median us/eval per round: " + " ".join(f"{x[0]:7.1f}" for x in r) + "   min: " + " ".join(f"{x[1]:7.1f}" for x in r) +
              f"   grad checksum {r[-1][2]:.9g}")
    full = res["merged"][-1][3]
    lo = res["loss-only"][-1][4]
    print("  loss sums equal (fused vs loss-only):", np.array_equal(full, lo), full, lo)
    del rep, eng
