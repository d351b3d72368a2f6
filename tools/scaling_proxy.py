#!/usr/bin/env python3
"""Strong-scaling PROXY for every BASELINE config on ONE MI355X (VERDICT r02 "Next round" item 2).

For N in {1, 2, 4, 8}: the engine evaluates rank 0's contiguous 1/N share of every term's point set (n_norm = the global N, exactly what
bench.py --gpus N installs on rank 0) and, for N > 1, completes the step with the engine's own RCCL all-reduce on a 1-rank communicator
(pinn_loss_grad_sharded_device), so the collective's launch and its double-precision loss exchange are inside the timed step; the result
is copied to the host every step, as in the bench.  What this does NOT contain is the xGMI hop of a real N-rank all-reduce (51 KB - 0.8 MB:
latency-bound, ~10-20 us on a ring of 8).  Projected speed-up = t(N = 1) / t(share of N).

    python tools/scaling_proxy.py [--configs cfg2 cfg3 cfg4 cfg5] [--out profiles/r03_scaling_proxy.json]

Per (config, N) it records the step time, the fused kernels' time (HIP events on sampled steps), the tiles per resident workgroup of the
largest launch, and names what bounds the share: "kernels" (tile work still dominates and shrinks with 1/N), "tile latency" (at most ~2
tiles per resident workgroup: one tile round costs the same whatever the point count) or "fixed cost" (pack + reduction + collective +
host turnaround are more than a third of the step)."""
import argparse, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--configs", nargs="*", default=["cfg2", "cfg3", "cfg4", "cfg5"])
ap.add_argument("--worlds", nargs="*", type=int, default=[1, 2, 4, 8])
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_scaling_proxy.json"))
args = ap.parse_args()
assert torch.cuda.is_available()
st = torch.cuda.current_stream()
results = []
for cfg in args.configs:
    wl = workloads.CONFIGS[cfg]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = [np.asarray(s) for s in (rep._state["pde_sets"] + rep._state["bc_sets"])]
    K, P = eng.K, eng.P
    theta = torch.tensor(np.asarray(rep.flat_init_params, dtype=np.float32), device="cuda")
    out_d = torch.zeros(P + K, dtype=torch.float32, device="cuda")
    out_h = torch.zeros(P + K, dtype=torch.float32).pin_memory()
    tw = rep._weights_now() if hasattr(rep, "_weights_now") else None
    tw = None if tw is None or np.all(np.asarray(tw) == 1.0) else list(np.asarray(tw, dtype=np.float32))
    have_comm = False
    base = None
    base_res = None
    theta0 = np.asarray(rep.flat_init_params, dtype=np.float32)
    for N in args.worlds:
        for k, s in enumerate(sets):
            n = s.shape[1]
            eng.set_points(k, s[:, : n // N] if N > 1 else s, n_norm=n)
        if N > 1 and not have_comm:
            eng.comm_init_rank(1, 0, npde.comm_unique_id())
            have_comm = True

        def step():
            if N > 1:
                eng.loss_grad_sharded_device(theta.data_ptr(), out_d.data_ptr(), tw, st.cuda_stream)
                out_h.copy_(out_d, non_blocking=True)
            else:
                eng.loss_grad_device(theta.data_ptr(), out_h.data_ptr(), tw, st.cuda_stream)
            st.synchronize()

        eng.set_timing(0, -1)
        t_one = time.perf_counter(); step(); step(); t_one = (time.perf_counter() - t_one) / 2
        nsteps = int(min(400, max(20, 0.25 / max(t_one, 1e-5))))
        for _ in range(max(10, nsteps // 4)):
            step()
        ts = []
        for _ in range(nsteps):
            t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        step_ms = float(np.median(ts)) * 1e3
        eng.set_timing(1, -1)
        km = []
        for _ in range(8):
            step()
            km.append(sum(g["ms"] for g in eng.group_timings() if g["ms"] > 0))
        eng.set_timing(0, -1)
        kernels_ms = float(np.median(km))
        desc = eng.describe()
        tiles = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"tiles=(\d+) blocks=(\d+)", desc)]
        tiles_per_wg = max(t / max(b, 1) for t, b in tiles)
        fixed_ms = step_ms - kernels_ms
        # the RESIDENT training loop on the same share (r04): evaluate -> in-stream all-reduce (N > 1) -> fused Adam + weight-image scatter,
        # no host synchronisation per iteration (pinn_adam_steps over the handle's communicator); time per iteration of one long call
        eng.adam_init(theta0)
        nres = int(min(2000, max(50, 0.5 / max(step_ms * 1e-3, 1e-5))))
        eng.adam(theta0, max(20, nres // 4), 1e-4, tw, init=False)
        t0 = time.perf_counter(); eng.adam(theta0, nres, 1e-4, tw, init=False); res_ms = (time.perf_counter() - t0) / nres * 1e3
        if base is None:
            base = step_ms
            base_res = res_ms
        bound = "fixed cost" if fixed_ms > step_ms / 3 else ("tile latency" if tiles_per_wg <= 2.0 else "kernels")
        r = {"config": wl.name, "world": N, "interior_points_share": sets[0].shape[1] // N, "step_ms": step_ms, "fused_kernels_ms": kernels_ms,
             "fixed_ms": fixed_ms, "tiles_per_resident_workgroup": tiles_per_wg, "projected_speedup": base / step_ms, "bound": bound,
             "launches": len([1 for l in desc.splitlines() if l.startswith("group")]),
             "resident_adam_ms_per_iteration": res_ms, "resident_projected_speedup": base_res / res_ms}
        results.append(r)
        print(f"{wl.name:44s} N={N}  step {step_ms:8.3f} ms  kernels {kernels_ms:8.3f}  fixed {fixed_ms:6.3f}  tiles/WG {tiles_per_wg:6.1f}  "
              f"projected speed-up {base / step_ms:5.2f}x  bound: {bound}   | resident Adam loop {res_ms:8.3f} ms/iteration, projected {base_res / res_ms:5.2f}x", flush=True)
    if have_comm:
        eng.comm_destroy()
    del rep, eng
with open(args.out, "w") as f:
    json.dump({"note": "single-GPU PROXY: rank 0's 1/N share + the engine's RCCL all-reduce on a 1-rank communicator; no xGMI hop measured",
               "results": results}, f, indent=1)
print("wrote", args.out)
