#!/usr/bin/env python3
"""bench.py — collocation-point residual+grad evals/sec on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

One "step" = one evaluation of the PINN loss (all K term losses) AND its gradient w.r.t. all network
weights for fixed theta and fixed collocation sets, delivered to the host (the reference's optimiser lives on
the host, src/discretize.jl:776-780), i.e. exactly what NeuralPDE.jl computes once per optimiser iteration
(src/discretize.jl:567-598 + Zygote, :778).

Workload at every N (config.workload; `--workload cfg3|cfg4|cfg5` selects another BASELINE config for profiling / scaling proxies, and
`--emulate-world N` times rank 0's share of an N-rank job on one GPU incl. the engine's RCCL call — tools/scaling_proxy.py):
BASELINE.json configs[1] — 2-D Poisson on the unit square, 4x64 tanh MLP,
QuasiRandomTraining: 65,536 interior points + 4 boundary terms x 65,536 points, all resident in HBM before
the timed region.  N > 1 is STRONG scaling (the same 65,536+4x65,536 points are sharded over the ranks in contiguous
blocks; one RCCL all-reduce of [gradient | per-term squared-residual sums] per step).
value = interior collocation points x steps / time  (the metric's unit: interior-point residual+grad evals/s;
the 4x65,536 boundary-term points ride along in every step and are counted in `point_terms_per_s`).

JSON extras: "roofline" (MFMA roofline of the dominant kernel = the fused residual kernel with the most device time, priced on the
matrix pipe that EXECUTES its hidden-layer products: in the default "split" GEMM mode every fp32 product is six bf16 MFMAs, so
`achieved` = executed bf16 MFMA flops / mean HIP-event duration over >= 10 launches sampled inside the timed region, `peak` = the dense
bf16 MFMA peak, `frac` <= 1 by construction; the fp32-equivalent rate against the fp32 MFMA peak — what BASELINE.json's north star
names — is kept as `frac_fp32_equiv`), "roofline_kernels" (the same for every fused kernel of the step) and "cpu_baseline" (the float64
oracle = CPU restatement of the reference algorithm, timed on this box's host cores on the same full-size workload at 32 / 64 / 96 / 128 threads,
the faster one reported; rank 0, N=1 only).  `value` keeps theta resident in HBM; `value_incl_theta_h2d` is SURVEY.md section 8d's
definition (theta crosses PCIe every step: the C-ABI host entry point).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md:41
PEAK_BF16_MFMA_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md:42 (dense; the 5 PF headline figure includes 2:1 sparsity)
SPLIT_PRODUCTS = 6                 # bf16 MFMAs per fp32-accurate product block (pinn_kernels2.hpp: mfma_split)


def split_mfma_flops(sizes, members, hp=None):
    """bf16 MFMA flops a split-GEMM launch EXECUTES: per tile and hidden->hidden layer the forward and dA GEMMs cover NG column groups of
    16, the dW GEMM (K = 32 points per MFMA) ceil(NG / 2) pairs of them; every fp32 product block is SPLIT_PRODUCTS bf16 MFMAs.
    Widths are the PADDED ones the kernel multiplies (64 / 128)."""
    hidden = sizes[1:-1]
    hp = hp or (64 if max(hidden) <= 64 else 128)
    nhh = len(hidden) - 1
    total = 0
    for h in members:
        ng = h["channels"] * h["pg"]
        cols = 16 * (2 * ng + 2 * ((ng + 1) // 2))
        total += h["tiles"] * nhh * 2 * hp * hp * cols * SPLIT_PRODUCTS
    return total


def algorithmic_flops_per_point(sizes, C):
    """flops(net, C) per point = 6*C*S - 2*C*n0*n1, S = sum_l n_{l-1} n_l."""
    S = sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
    return 6 * C * S - 2 * C * sizes[0] * sizes[1]


CPU_THREADS = (32, 64, 96, 128)   # intra-op thread counts of the CPU baseline (capped by the box's hardware threads): torch's CPU GEMMs stop scaling
                                   # somewhere in this range on the 256-thread hosts and COLLAPSE at every hardware thread (r04: 43 pts/s), so "all" is not tried


def cpu_baseline(npde, wl, sets, nevals=5, budget_s=9.0, chunk=16384):
    """Time the float64 oracle (stencil mode = the reference's algorithm: 6 batched forward passes per Poisson residual + reverse
    mode) on this box's host cores on the SAME workload as the GPU leg — all 65,536 interior + 4 x 65,536 boundary points, evaluated
    in chunks of `chunk` points per term to bound memory — at 32 / 64 / 96 / 128 intra-op threads (as far as the box has them; about 10 s of
    work per leg); value = interior points / median eval time of the FASTEST setting, every leg is stated."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import pinn_oracle as po
    import helpers
    ncpu = os.cpu_count() or 1
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    N = [s.shape[1] for s in sets]
    nchunks = max((n + chunk - 1) // chunk for n in N)

    def one(csize=chunk, limit=None):
        t = time.perf_counter()
        nc = max((n + csize - 1) // csize for n in N)
        for c in range(nc if limit is None else min(nc, limit)):
            part, w = [], []
            for k, s_ in enumerate(sets):
                lo, hi = c * csize, min((c + 1) * csize, N[k])
                part.append(s_[:, lo:hi] if lo < hi else s_[:, :1])
                w.append((hi - lo) / N[k] if lo < hi else 0.0)
            po.loss_and_grad(prob, wl.theta, part, weights=w, mode="stencil")
        return time.perf_counter() - t

    runs = []
    probe = 2048                                # points per term of the probe that decides whether a thread count gets full evaluations
    for nt in sorted(set(min(t, ncpu) for t in CPU_THREADS)):
        torch.set_num_threads(nt)
        one(probe, 1)                           # warm-up (thread pool, allocator) on the probe ...
        est = one(probe, 1) * (max(N) / probe)  # ... and an estimate of one full evaluation from it
        if est > 20.0:
            # this thread count is far off (oversubscribed hosts: torch's CPU GEMMs collapse at every hardware thread): keep the bounded
            # sample as its figure instead of spending minutes on full evaluations (the contract: about 10-30 s of CPU work per leg)
            runs.append({"threads": nt, "evals": 0, "median_s": float(est), "min_s": float(est), "max_s": float(est), "value": N[0] / float(est),
                         "sample": f"one chunk of {probe} points per term, extrapolated to the full workload"})
            continue
        one()
        times, t0 = [], time.perf_counter()
        while len(times) < nevals and (time.perf_counter() - t0 < budget_s or len(times) < 3):
            times.append(one())
        runs.append({"threads": nt, "evals": len(times), "median_s": float(np.median(times)), "min_s": float(min(times)), "max_s": float(max(times)),
                     "value": N[0] / float(np.median(times)), "sample": "full workload"})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "interior-point residual+grad evals/s", "cores": best["threads"], "kind": "port",
            "host_cpus": ncpu, "evals": best["evals"], "median_s": best["median_s"], "min_s": best["min_s"], "max_s": best["max_s"],
            "thread_counts_tried": runs,
            "sample": f"median of {best['evals']} evals of the float64 stencil-mode oracle (torch CPU) on the full workload: {N[0]} interior + "
                      f"{len(N) - 1}x{N[1]} boundary points in chunks of {chunk}; timed with " +
                      ", ".join(f"{r['threads']} threads ({r['value']:.3g} pts/s, {r['sample']})" for r in runs) + f" of {ncpu} hardware threads, the fastest "
                      f"one reported; Julia/NeuralPDE.jl itself is not installable here (no network)"}


def pmc_traffic(kernel_key, points):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (tools/pmc_profile.sh writes
    profiles/pmc_traffic.json: 2 x FETCH_SIZE [gfx950 wide-read correction] + WRITE_SIZE, KB -> bytes, per kernel and workload size).
    Counters cannot be collected inside this process, so the line carries the profile's figure together with its source file, or
    null when no profile matches this kernel / point count."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None, None
    for e in tab.get("kernels", []):
        if e.get("key") == kernel_key and e.get("points_per_launch") == points:
            return e.get("bytes_per_launch"), e.get("source")
    return None, None


def kernel_facts(kernel_key, points):
    """Counter / ISA facts of a kernel from the committed profile summaries (profiles/kernel_facts.json, written by tools/pmc_summarize.py
    and tools/isa_report.py from the rocprofv3 --pmc passes and the disassembly of the shipped objects): matrix-pipe busy fraction
    (SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / (GRBM_GUI_ACTIVE / XCDs)) and VALU instructions per interior tile.  Not measurable inside this
    process; null when no committed profile matches this kernel and size."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_facts.json")) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return {}
    for e in tab.get("kernels", []):
        if e.get("key") == kernel_key and e.get("points_per_launch") in (None, points):
            return e
    return {}


def roofline_entries(eng, kern_ms, sizes, world):
    """One roofline entry per fused residual LAUNCH of the step.  A MERGED launch (the interior jet set and the value-only boundary set
    of one network walked by one persistent kernel, pinn_group_launched_by) is one entry covering both groups' points and flops.
    The entry is priced on the matrix pipe that executes the launch's hidden-layer products:
      * gemm = split-bf16 (default, 64- / 128-wide kernels): every fp32 product block is six v_mfma_f32_16x16x32_bf16, so `achieved` =
        executed bf16 MFMA flops / mean kernel time, `peak` = the dense bf16 MFMA peak (2,500 TF/s), `frac` = their ratio (<= 1 by
        construction: the pipe cannot execute more than its peak);
      * gemm = fp32 (pinn_set_option / narrower nets): `achieved` = executed fp32 flops (SURVEY.md section 8d formula with the channels the
        kernel carries) / time against the fp32 MFMA peak (157.3 TF/s).
    `frac_fp32_equiv` is, in both cases, the fp32-equivalent executed flops (6 C S - 2 C n0 n1 per point with C = the jet channels each
    member carries) / time / fp32 MFMA peak — the figure BASELINE.json's north star names ("% of the fp32 MFMA roofline on the layer
    GEMMs"); with split products it may exceed what the fp32 pipe alone could do, which is why it is not `frac`.  The 2-D Poisson interior
    residual as written needs C = 5 channels (u, u_x, u_y, u_xx, u_yy: 373,120 flop/point, the section 8d figure); the kernel carries
    u_xx + u_yy as ONE forward-Laplacian channel (C = 4), so the section 8d "useful work" rate is reported separately as
    frac_algorithmic."""
    import re
    import numpy as np
    groups = eng.group_timings()
    desc = eng.describe()
    names = dict((int(m.group(1)), m.group(2)) for m in re.finditer(r"group (\d+).*?kernel=(\S+)", desc))
    coupled = set(int(m.group(1)) for m in re.finditer(r"group (\d+) \[coupled", desc))
    m = re.search(r"gemm=(split-bf16\S+)", desc)
    split = m.group(1) if m else None
    per_kernel = []
    for gi, g in enumerate(groups):
        if g["launched_by"] != gi:
            continue                                    # rode on another group's launch
        members = [h for h in groups if h["launched_by"] == gi]
        ms = float(np.mean(kern_ms[:, gi]))
        if not (ms > 0):
            continue
        f_exec = sum(algorithmic_flops_per_point(sizes, h["channels"]) * h["points"] for h in members)
        # §8d algorithmic channel count: the forward-Laplacian channel stands for the pure second derivatives it sums (2-D: +1, 3-D: +2)
        def lap_extra(key):                              # kernel names carry the Laplacian axis mask as _L<mask>_
            m = re.search(r"_L(\d+)_", key)
            return max(0, bin(int(m.group(1))).count("1") - 1) if m else 0
        f_alg = sum(algorithmic_flops_per_point(sizes, h["channels"] + lap_extra(names.get(h["group"], ""))) * h["points"] for h in members)
        pts = sum(h["points"] for h in members)
        key = "+".join(names.get(h["group"], f"group{h['group']}") for h in members)
        traffic, tsrc = pmc_traffic(key, pts) if world == 1 else (None, None)
        facts = kernel_facts(key, pts) if world == 1 else {}
        kind = "coupled reverse launch (forward launches not timed)" if gi in coupled else ("merged interior+boundary residual+grad" if len(members) > 1 else
               ("interior residual+grad" if g["channels"] > 1 else "boundary residual+grad"))
        tf_exec, tf_alg = f_exec / (ms * 1e-3) / 1e12, f_alg / (ms * 1e-3) / 1e12
        fam2 = all(names.get(h["group"], "").startswith("F2_") for h in members)
        hp = int(re.search(r"_HP(\d+)_", names.get(gi, "_HP0_")).group(1))
        on_bf16 = bool(split) and fam2 and hp in (64, 128) and "dW" in split
        entry = {"traffic": traffic, "traffic_source": tsrc,
                 "kernel": f"{'k_wave2m' if len(members) > 1 else 'k_wave2'}<{key}, FUSED> ({kind}, neuron-split workgroups)",
                 "kernel_ms": ms, "kernel_ms_min": float(np.min(kern_ms[:, gi])), "kernel_ms_max": float(np.max(kern_ms[:, gi])),
                 "launches_sampled": int(kern_ms.shape[0]), "points_per_launch": pts,
                 "executed_channels": [h["channels"] for h in members], "executed_flops_per_launch": f_exec, "algorithmic_flops_per_launch": f_alg,
                 "achieved_fp32_equiv": tf_exec, "frac_fp32_equiv": tf_exec / PEAK_FP32_MFMA_TFLOPS,
                 "achieved_algorithmic": tf_alg, "frac_algorithmic": tf_alg / PEAK_FP32_MFMA_TFLOPS,
                 "algorithmic_bytes_per_launch": 4 * sizes[0] * pts,
                 "mfma_busy": facts.get("mfma_busy"), "valu_insts_per_tile": facts.get("valu_insts_per_tile"), "facts_source": facts.get("source")}
        if on_bf16:
            for h in members:
                h["pg"] = int(re.search(r"_PG(\d+)\(", names.get(h["group"], "_PG1(")).group(1))
            f_bf = split_mfma_flops(sizes, members, hp)
            tf_bf = f_bf / (ms * 1e-3) / 1e12
            entry.update({"bound": "mfma-bf16(split x6)", "achieved": tf_bf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf_bf / PEAK_BF16_MFMA_TFLOPS,
                          "gemm": split, "executed_bf16_mfma_flops_per_launch": f_bf})
        else:
            entry.update({"bound": "mfma-fp32", "achieved": tf_exec, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf_exec / PEAK_FP32_MFMA_TFLOPS,
                          "gemm": split if (split and fam2) else "fp32"})
        per_kernel.append(entry)
    return per_kernel


def bench_f64_sharded(args, eng, rep, sets, wl, npde, world, rank):
    """--precision f64 with --gpus N > 1 (or --emulate-world N: rank 0's share of an N-rank job over a 1-rank RCCL communicator): every
    rank evaluates its contiguous shard of every set with the double kernels, then the ENGINE's communicator all-reduces [P + K] doubles
    (pinn_loss_grad_sharded_device_f64, ncclDouble on the evaluation's stream); [gradient | sums] copied to pinned host memory every step
    as in the fp32 line.  Strong scaling: the global sets are fixed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    eworld = args.emulate_world if args.emulate_world > 0 else world
    erank = 0 if args.emulate_world > 0 else rank
    n_glob = [s_.shape[1] for s_ in sets]
    for k, s_ in enumerate(sets):
        n = s_.shape[1]
        lo, hi = (n * erank) // eworld, (n * (erank + 1)) // eworld
        eng.set_points_f64(k, np.asarray(s_, dtype=np.float64)[:, lo:hi], n_norm=n)
    if world > 1:
        uid = [npde.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init_rank(world, rank, uid[0])
    else:
        eng.comm_init_rank(1, 0, npde.comm_unique_id())
    K, P = eng.K, eng.P
    th = np.asarray(rep.flat_init_params, dtype=np.float64)
    d_th = torch.tensor(th, dtype=torch.float64, device="cuda")
    d_out = torch.zeros(P + K, dtype=torch.float64, device="cuda")
    h_out = torch.zeros(P + K, dtype=torch.float64).pin_memory()
    stream = torch.cuda.current_stream()

    def step():
        eng.loss_grad_sharded_device_f64(d_th.data_ptr(), d_out.data_ptr(), None, stream.cuda_stream)
        h_out.copy_(d_out, non_blocking=True)
        stream.synchronize()

    for _ in range(max(2, args.warmup)):
        step()
    first = h_out.clone()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.equal(first, h_out), "float64 sharded evaluation is not reproducible"
    ms = dt / args.steps * 1e3
    path = eng.get_option("f64_path")
    eng.comm_destroy()
    return {"metric": "collocation-point residual+grad evals/sec, 2D Poisson 4x64 MLP (float64 evaluation mode)" if args.workload == "cfg2" else f"collocation-point residual+grad evals/sec ({args.workload}, float64 evaluation mode)",
            "value": n_glob[0] / (ms * 1e-3), "unit": "interior-point residual+grad evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name if hasattr(wl, "name") else args.workload, "points_per_term_global": n_glob, "precision": "f64", "f64_path": path,
                       "emulate_world": args.emulate_world or None,
                       "parallelism": (f"PROXY: rank 0's 1/{args.emulate_world} share on one GPU, 1-rank RCCL communicator (projected whole-job rate, no xGMI hop measured)"
                                       if args.emulate_world > 0 else f"dp{world}: contiguous shards of every set, one ncclDouble all-reduce per step"),
                       "entry": "pinn_loss_grad_sharded_device_f64 (shard evaluation in double + the engine's ncclDouble all-reduce of [P + K]) + D2H to pinned memory, one synchronisation per step"},
            "proxy": args.emulate_world > 0,
            "loss": float((h_out[P:].numpy() / np.array(n_glob)).sum()) if args.emulate_world == 0 else None}


F64_MFMA_PEAK_TF = 78.6          # dense v_mfma_f64_16x16x4_f64 peak of MI355X (MI355X_MICROARCH.md; = the f64 vector peak)


def bench_f64(args, eng, rep, sets, wl, npde=None, world=1, rank=0):
    """One line for the FLOAT64 evaluation mode on the same workload: a step = pinn_loss_grad_f64 (theta from host memory in double, loss +
    gradient back in double; the chunked tile / small-entry / weight-gradient / reduce kernels of csrc/pinn_kernels5.hpp + pinn_kernels4.hpp).
    roofline: the hidden-layer GEMM flops the f64 MFMAs execute (forward + dA in the tile kernel, dW in the weight-gradient kernel: 3 x 2 x
    H^2 per hidden-to-hidden layer, point and jet channel — the channel counts are the FLOAT64 kernels' (Poisson interior: C = 5, no
    forward-Laplacian channel)) over the WHOLE evaluation's wall time against the dense f64 MFMA peak; the per-kernel durations are in
    profiles/r05_f64_kernel_stats.txt."""
    import numpy as np
    import torch
    eng.set_option("precision", "f64")
    if world > 1 or args.emulate_world > 0:
        return bench_f64_sharded(args, eng, rep, sets, wl, npde, world, rank)
    for k, s_ in enumerate(sets):
        eng.set_points_f64(k, s_)
    th = np.asarray(rep.flat_init_params, dtype=np.float64)
    K = eng.K
    w = np.ones(K)
    for _ in range(max(2, args.warmup)):
        l0, g0 = eng.loss_grad_f64(th, w)
    path = eng.get_option("f64_path")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l1, g1 = eng.loss_grad_f64(th, w)
    torch.cuda.synchronize()
    dt_host = time.perf_counter() - t0
    assert np.array_equal(l1, l0) and np.array_equal(g1, g0), "float64 evaluation is not reproducible"
    # `value`: as the fp32 line — theta and the result resident in HBM (pinn_loss_grad_device_f64, double device buffers), every step
    # synchronised; the host-entry rate (theta H2D + gradient D2H in double inside the step) rides along as value_host_entry
    P = eng.P
    d_th = torch.tensor(th, dtype=torch.float64, device="cuda")
    d_out = torch.zeros(P + K, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(max(2, args.warmup)):
        eng.loss_grad_device_f64(d_th.data_ptr(), d_out.data_ptr(), w, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.loss_grad_device_f64(d_th.data_ptr(), d_out.data_ptr(), w, st)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = d_out.cpu().numpy()
    assert np.array_equal(out[:P], g0) and np.allclose(out[P:] / np.array([s_.shape[1] for s_ in sets]), l0, rtol=1e-15, atol=0), "device-pointer and host entry disagree"
    ms = dt / args.steps * 1e3
    ms_host = dt_host / args.steps * 1e3
    sizes = list(wl.chains[0].sizes)
    hh = sum(sizes[i] * sizes[i + 1] for i in range(1, len(sizes) - 2))          # hidden-to-hidden products per point and channel
    chans = [int(c) for c in eng.describe().split("f64_channels=")[1].split()[0].split(",")] if "f64_channels=" in eng.describe() else None
    n = [s_.shape[1] for s_ in sets]
    if chans is None:
        chans = [5] + [1] * (K - 1) if args.workload == "cfg2" else [1] * K
    flops = sum(3 * 2 * hh * c * nn for c, nn in zip(chans, n))
    ach = flops / (ms * 1e-3) / 1e12
    # HBM-side bytes of one evaluation from the committed PMC passes (profiles/pmc_traffic_f64.json: this workload at 65,536 points per term)
    traffic, tsrc, busy = None, None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_f64.json")) as f:
            tj = json.load(f)
        if args.workload == "cfg2" and all(nn == 65536 for nn in n) and path == "mfma":
            traffic, tsrc = tj["bytes_per_evaluation"], tj["source"]
            busy = {k_["kernel"].split("(")[0].replace("void pk::", ""): round(k_["mfma_busy"], 3) for k_ in tj["kernels"] if k_["mfma_busy"] > 0}
    except (OSError, KeyError, ValueError):
        pass
    return {"metric": "collocation-point residual+grad evals/sec, 2D Poisson 4x64 MLP (float64 evaluation mode)" if args.workload == "cfg2" else f"collocation-point residual+grad evals/sec ({args.workload}, float64 evaluation mode)",
            "value": n[0] / (ms * 1e-3), "unit": "interior-point residual+grad evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name if hasattr(wl, "name") else args.workload, "points_per_term": n, "precision": "f64", "f64_path": path,
                       "entry": "pinn_loss_grad_device_f64 (theta and [gradient | sums] resident in HBM as doubles, one synchronisation per step)"},
            "value_host_entry": n[0] / (ms_host * 1e-3), "ms_per_step_host_entry": ms_host,          # pinn_loss_grad_f64: theta H2D + gradient D2H inside the step
            "roofline": {"bound": "mfma-f64", "achieved": ach, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / F64_MFMA_PEAK_TF, "traffic": traffic,
                         "traffic_source": tsrc, "mfma_busy_per_kernel": busy,
                         "flops_per_step": flops, "channels_per_term": chans,
                         "note": "executed f64 MFMA flops (forward + dA + dW hidden-layer GEMMs) / whole-evaluation wall time; kernel-level durations: profiles/r05_f64_kernel_stats.txt"},
            "loss_terms": [float(x) for x in l1]}



def emit(line):
    """the ONE JSON line, as the LAST line of stdout: native libraries (RCCL prints a version banner through C stdio when a communicator is made) keep
    their text in C buffers that would otherwise be flushed at exit, i.e. behind the JSON"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2",
                    help="BASELINE.json config; cfg2 (default) is the one the metric is quoted on — the others are scaling-proxy / profiling runs")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="scaling PROXY on one GPU: this process evaluates rank 0's contiguous 1/N share of every term's point set (n_norm = "
                         "the global N) and runs the engine's RCCL all-reduce on a 1-rank communicator, so the collective's launch cost is "
                         "inside the step; value = what N such ranks would deliver together (tools/scaling_proxy.py, profiles/r03_scaling_proxy.json)")
    ap.add_argument("--gemm", choices=["split", "fp32"], default="split",
                    help="GEMM arithmetic of the 64- / 128-wide kernels (pinn_set_option): split = three-piece bf16 split products (default, the "
                         "product), fp32 = v_mfma_f32_16x16x4_f32")
    ap.add_argument("--resident", action="store_true",
                    help="time the RESIDENT training loop instead of the evaluation: one step = one Adam iteration with theta, moments and point "
                         "sets in HBM (pinn_adam_steps; N > 1 / --emulate-world: evaluate -> in-stream all-reduce -> fused update on every rank, no host "
                         "synchronisation inside the loop); the K timed steps are ONE call.  Not the headline metric (which delivers loss + "
                         "gradient to a host optimiser every step) — the loop a training run actually executes")
    ap.add_argument("--precision", choices=["f32", "f64"], default="f32",
                    help="f64: the engine's FLOAT64 evaluation mode (pinn_set_option(h, \"precision\", \"f64\"): the reference's default eltype, "
                         "src/discretize.jl:432-449) on the same workload — its own line with a roofline entry against the f64 MFMA peak (r05: "
                         "v_mfma_f64_16x16x4_f64 tile kernels, csrc/pinn_kernels5.hpp).  Not the headline metric (north star: fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events", choices=["all", "none"], default="all",
                    help="HIP events recorded inside the timed region around every fused residual kernel (sampled steps only), or none")
    ap.add_argument("--event-every", type=int, default=0, help="record the HIP events on every M-th timed step (a start/stop pair plus "
                    "its read-back costs ~10 us of host+dispatch time per kernel; the sampled launches are inside the timed region). "
                    "0 (default): steps // 10, i.e. at least 10 sampled launches of every kernel")
    args = ap.parse_args()

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's peer-buffer exchange fails with hipIpcGetMemHandle errors
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert not (args.emulate_world and world > 1), "--emulate-world is a single-GPU proxy"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Control plane (rendezvous, barrier, max-over-ranks of the elapsed time): a gloo process group over TCP.  Data plane (the ONE
    # all-reduce per step of [gradient | per-term sums]): the ENGINE's own RCCL communicator (pinn_comm_init_rank +
    # pinn_loss_grad_sharded_device, issued on the evaluation's stream) — torch.distributed carries no tensor of the timed region.
    # PINN_BENCH_COMM=torch falls back to a torch.distributed all-reduce (nccl = RCCL, or gloo with PINN_BENCH_BACKEND=gloo and the
    # ranks folded onto the visible devices: functional check of the N > 1 path on a 1-GPU box).
    backend = os.environ.get("PINN_BENCH_BACKEND", "nccl")
    comm_mode = os.environ.get("PINN_BENCH_COMM", "engine" if backend == "nccl" else "torch")
    local_dev = local_rank % torch.cuda.device_count()        # (== local_rank whenever the node has one GPU per rank)
    torch.cuda.set_device(local_dev)
    data_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if comm_mode == "torch" and backend == "nccl":
            data_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_dev))

    import pinn_import
    npde = pinn_import.load()
    from neuralpde_jl_amd import workloads

    wl = workloads.cfg2_poisson2d(points=args.points) if args.workload == "cfg2" else workloads.CONFIGS[args.workload]()
    disc = wl.discretization()
    rep = npde.symbolic_discretize(wl.pde_system, disc)
    eng = rep.engine
    assert eng.L.backend == "hip"
    if args.gemm != "split":
        eng.set_option("gemm", args.gemm)
    sets = rep.pde_train_sets + rep.bcs_train_sets
    K, P = eng.K, eng.P
    if args.precision == "f64":
        assert not args.resident, "--precision f64: the evaluation line (the resident double loop is timed by tools/time_f64.py)"
        line = bench_f64(args, eng, rep, sets, wl, npde, world, rank)
        if rank == 0:
            emit(line)
        if world > 1:
            dist.destroy_process_group()
        return
    n_glob = [s.shape[1] for s in sets]
    theta0 = np.asarray(rep.flat_init_params, dtype=np.float32)
    tw = rep._weights_now() if hasattr(rep, "_weights_now") else None        # cfg4: bc weights 10 (NonAdaptiveLoss)
    tw = None if tw is None or np.all(np.asarray(tw) == 1.0) else list(np.asarray(tw, dtype=np.float32))
    # shard every term's set into contiguous column blocks (strong scaling); --emulate-world: rank 0's share of an N-rank job
    eworld = args.emulate_world if args.emulate_world > 0 else world
    erank = 0 if args.emulate_world > 0 else rank
    if eworld > 1:
        for k, s in enumerate(sets):
            n = s.shape[1]
            lo, hi = (n * erank) // eworld, (n * (erank + 1)) // eworld
            eng.set_points(k, s[:, lo:hi], n_norm=n)
    theta_d = torch.tensor(theta0, dtype=torch.float32, device="cuda")
    out_d = torch.zeros(P + K, dtype=torch.float32, device="cuda")
    out_h = torch.zeros(P + K, dtype=torch.float32).pin_memory()
    stream = torch.cuda.current_stream()

    # N = 1: the reduction kernel writes [grad | sums] straight into the pinned (device-mapped, coherent) host buffer — no
    # copy command; N > 1: the all-reduce needs the vector in HBM first, then one D2H copy.
    if world > 1 and comm_mode == "engine":
        uid = [npde.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            eng.comm_init_rank(world, rank, uid[0])
        except Exception as e:                                    # keep the run alive: report which path carried the all-reduce
            print(f"[bench rank {rank}] engine communicator failed ({e}); falling back to torch.distributed", file=sys.stderr)
            comm_mode = "torch-fallback"
        modes = [None] * world
        dist.all_gather_object(modes, comm_mode)
        if any(m != "engine" for m in modes):
            if comm_mode == "engine":
                eng.comm_destroy()
            comm_mode = "torch-fallback"
            # second fallback: if torch's own RCCL group cannot be formed or fails its first collective either, the step reduces on the
            # host through the gloo control group (slow, but the run reports a number and says which path carried it)
            ok = True
            try:
                data_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_dev))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe, group=data_group)
                torch.cuda.synchronize()
                ok = abs(float(probe.item()) - world) < 0.5
            except Exception as e:
                print(f"[bench rank {rank}] torch.distributed nccl group failed ({e}); reducing on the host over gloo", file=sys.stderr)
                ok = False
            oks = [None] * world
            dist.all_gather_object(oks, ok)
            if not all(oks):
                data_group = None
                comm_mode = "gloo-host-fallback"
    if args.emulate_world > 0:
        eng.comm_init_rank(1, 0, npde.comm_unique_id())           # the real RCCL entry points on a 1-rank communicator
    sharded = world > 1 or args.emulate_world > 0

    def step():
        if sharded and (comm_mode == "engine" or args.emulate_world > 0):
            eng.loss_grad_sharded_device(theta_d.data_ptr(), out_d.data_ptr(), tw, stream.cuda_stream)
            out_h.copy_(out_d, non_blocking=True)
        elif world > 1:
            eng.loss_grad_device(theta_d.data_ptr(), out_d.data_ptr(), tw, stream.cuda_stream)
            if data_group is not None:
                dist.all_reduce(out_d, group=data_group)
            else:                                                  # gloo functional check: reduce on the host
                tmp = out_d.cpu()
                dist.all_reduce(tmp)
                out_d.copy_(tmp)
            out_h.copy_(out_d, non_blocking=True)
        else:
            eng.loss_grad_device(theta_d.data_ptr(), out_h.data_ptr(), tw, stream.cuda_stream)
        stream.synchronize()                 # the optimiser needs loss + gradient on the host every iteration

    if args.resident:
        assert comm_mode == "engine" or world == 1, "--resident needs the engine-owned communicator"
        eng.adam_init(theta0)

        def run_steps(n):
            eng.adam(theta0, n, 1e-3, tw, init=False)          # pinn_adam_steps: n iterations, one host synchronisation at the end

    host_path_ms = None
    loss_only_ms = None
    if args.resident:
        run_steps(30 + args.warmup)
    elif world == 1 and not sharded:
        # (measured BEFORE the warm-up and the timed region: it doubles as the clock / cache warm-up of the device)
        # cross-check of the zero-copy delivery against a plain device-buffer evaluation + copy
        eng.set_timing(0, -1)
        step()
        eng.loss_grad_device(theta_d.data_ptr(), out_d.data_ptr(), tw, stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(out_d.cpu().numpy(), out_h.numpy()), "zero-copy host delivery differs from the device buffer"
        # the C-ABI host entry point (theta from host memory, loss + gradient back to host memory: PCIe both ways)
        th = np.ascontiguousarray(theta0, dtype=np.float32)
        for _ in range(5):
            eng.loss_grad(th, tw)
        nh = 25
        t1 = time.perf_counter()
        for _ in range(nh):
            eng.loss_grad(th, tw)
        host_path_ms = (time.perf_counter() - t1) / nh * 1e3
        # the loss-only evaluation through the same entry point (grad = NULL: no reverse sweep), same numbers for the losses
        l_full, _ = eng.loss_grad(th, tw)
        l_only, _ = eng.loss_grad(th, tw, want_grad=False)
        assert np.allclose(l_only, l_full, rtol=1e-13, atol=0), "loss-only evaluation differs from the fused evaluation's losses"
        t1 = time.perf_counter()
        for _ in range(nh):
            eng.loss_grad(th, tw, want_grad=False)
        loss_only_ms = (time.perf_counter() - t1) / nh * 1e3
    else:
        for _ in range(30):                  # N > 1: the same device clock / cache settling the N = 1 leg gets from the checks above
            step()
    for _ in range(0 if args.resident else args.warmup):
        step()
    ev_level, ev_group = {"all": 1, "none": 0}[args.events], -1
    if args.resident:
        ev_level = 0                                          # (no per-step host turn to read events back in)
    eng.set_timing(ev_level, ev_group)
    if not args.resident:
        step()
    kern_ms = []
    every = max(1, args.event_every if args.event_every > 0 else args.steps // 10)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.resident:
        run_steps(args.steps)
    for i in range(0 if args.resident else args.steps):
        sampled = ev_level > 0 and i % every == 0
        if ev_level > 0 and every > 1:
            eng.set_timing(ev_level if sampled else 0, ev_group)
        step()
        if sampled:
            kern_ms.append([g["ms"] for g in eng.group_timings()])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        if args.resident:                                     # the loss terms at the loop's final theta (outside the timed region)
            eng.set_timing(0, -1)
            step()
        res = out_h.numpy()
        losses = res[P:] / np.array(n_glob)
        ngroups = len(eng.group_timings())
        kern_ms = np.array(kern_ms) if kern_ms else np.full((1, ngroups), np.nan)
        sizes = wl.chains[0].sizes
        n_int = n_glob[0]
        per_kernel = roofline_entries(eng, kern_ms, sizes, world if not args.emulate_world else 2)
        line = {
            "metric": "collocation-point residual+grad evals/sec, 2D Poisson 4x64 MLP" if args.workload == "cfg2" else
                      f"collocation-point residual+grad evals/sec, {wl.name}",
            "value": n_int * args.steps / el,
            "unit": "interior-point residual+grad evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if "gemm=split" not in eng.describe() else
                     ("f32, hidden-layer forward / dA / dW GEMMs as 3-piece bf16 split products (6 bf16 MFMAs per product block, fp32 accumulation; "
                      "24 mantissa bits rebuilt; the five small piece products on an accumulator of their own: zero-mean, a third of the rms error "
                      "of an fp32-MFMA fmaf chain per GEMM output, tools/micro/split_bias_probe.hip) — gradient rel. L2 ~3e-8 of the float64 mode at "
                      "the goldens' glorot parameters, 0.3-1.1x a plain float32 evaluation's error at trained parameters (DESIGN.md section 6.1); "
                      "first / last layer, activations, reductions in fp32; --gemm fp32 runs the exact fp32-MFMA kernels"),
            "data": "synthetic",
            "config": {"workload": wl.name, "interior_points": n_int, "boundary_terms": K - len(rep.pde_train_sets),
                       "boundary_points_per_term": n_glob[-1], "theta": P,
                       "parallelism": f"point-shard x{world}" if world > 1 else "single",
                       "all_reduce": ({"engine": "engine-owned RCCL communicator (pinn_loss_grad_sharded_device)", "torch": f"torch.distributed ({backend})",
                                       "torch-fallback": "torch.distributed (nccl) after the engine communicator failed",
                                       "gloo-host-fallback": "host reduction over gloo after both RCCL paths failed"}[comm_mode] if world > 1 else None)},
            "point_terms_per_s": sum(n_glob) * args.steps / el,
            # SURVEY.md section 8d's definition of the metric has theta cross PCIe every step: the C-ABI host entry point below
            "value_incl_theta_h2d": (n_int / (host_path_ms * 1e-3)) if host_path_ms else None,
            "host_entry_ms_per_step": host_path_ms,     # pinn_loss_grad: theta host -> device, results device -> host (PCIe-inclusive)
            "loss_only_host_entry_ms": loss_only_ms,    # pinn_loss_grad(grad = NULL): the loss-only evaluation through the same entry point
            "loss_terms": [float(v) for v in losses],
        }
        if args.resident:
            line["config"]["loop"] = ("resident Adam: theta / moments / point sets in HBM, " + ("evaluate -> in-stream all-reduce -> fused update on every rank, "
                                      if sharded else "evaluate -> fused update, ") + f"{args.steps} iterations per host call (pinn_adam_steps), no per-step host synchronisation")
            line["metric"] += " (resident training loop incl. the optimiser step)"
            line["loss_terms_note"] = "at the loop's final theta"
        if args.emulate_world > 0:
            line["config"].update({"emulate_world": args.emulate_world, "parallelism": f"PROXY: rank 0's 1/{args.emulate_world} share on one GPU",
                                   "all_reduce": "engine-owned RCCL all-reduce on a 1-rank communicator (launch cost inside the step)"})
            line["proxy"] = True
            line["value_note"] = ("projected whole-job rate if all N ranks take as long as this share (no xGMI hop measured): "
                                  "global interior points x steps / time of rank 0's share")
        if per_kernel:
            dom = int(np.argmax([k["kernel_ms"] for k in per_kernel]))          # dominant = the launch with the most device time
            all_ms = float(sum(k["kernel_ms"] for k in per_kernel))
            flops_all = sum(k["executed_flops_per_launch"] for k in per_kernel)
            roof = dict(per_kernel[dom])
            roof.update({"all_fused_kernels_ms": all_ms, "all_fused_kernels_tflops_fp32_equiv": flops_all / (all_ms * 1e-3) / 1e12,
                         "all_fused_kernels_frac_fp32_equiv": flops_all / (all_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "events": f"HIP events around every fused kernel on every {every}. step of the timed region: {kern_ms.shape[0]} launches averaged",
                         "note": ("frac = achieved / peak on the pipe named in `bound`: for split-bf16 kernels the bf16 MFMA flops the launch executes (six "
                                  "MFMAs per fp32 product block) against the dense bf16 peak of 2,500 TF/s; frac_fp32_equiv = the fp32-equivalent executed flops "
                                  "against the fp32 MFMA peak of 157.3 TF/s (BASELINE.json's north-star figure; not a fraction of any one pipe for split kernels); "
                                  "peaks at the 2.4 GHz peak clock, the shader clock under this load is ~1.9-2.0 GHz; traffic = HBM-side bytes per launch from the "
                                  "committed rocprofv3 --pmc passes named in traffic_source, mfma_busy / valu_insts_per_tile from the profile named in facts_source "
                                  "(null: no committed profile for this kernel and size)")})
            line["roofline"] = roof
            line["roofline_kernels"] = per_kernel
        if world == 1 and not sharded and not args.no_cpu_baseline and args.workload == "cfg2":
            line["cpu_baseline"] = cpu_baseline(npde, wl, sets)
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
