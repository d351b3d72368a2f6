#!/usr/bin/env python3
"""bench.py — collocation-point residual+grad evals/sec on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

One "step" = one evaluation of the PINN loss (all K term losses) AND its gradient w.r.t. all network
weights for fixed theta and fixed collocation sets, delivered to the host (the reference's optimiser lives on
the host, src/discretize.jl:776-780), i.e. exactly what NeuralPDE.jl computes once per optimiser iteration
(src/discretize.jl:567-598 + Zygote, :778).

Workload at every N (config.workload): BASELINE.json configs[1] — 2-D Poisson on the unit square, 4x64 tanh MLP,
QuasiRandomTraining: 65,536 interior points + 4 boundary terms x 65,536 points, all resident in HBM before
the timed region.  N > 1 is STRONG scaling (the same 65,536+4x65,536 points are sharded over the ranks in contiguous
blocks; one RCCL all-reduce of [gradient | per-term squared-residual sums] per step).
value = interior collocation points x steps / time  (the metric's unit: interior-point residual+grad evals/s;
the 4x65,536 boundary-term points ride along in every step and are counted in `point_terms_per_s`).

JSON extras: "roofline" (fp32 MFMA roofline of the dominant kernel = the fused interior residual kernel,
algorithmic flops per SURVEY.md §8d / DESIGN.md ÷ its HIP-event duration) and "cpu_baseline" (the float64 oracle =
CPU restatement of the reference algorithm, timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md:41


def algorithmic_flops_per_point(sizes, C):
    """flops(net, C) per point = 6*C*S - 2*C*n0*n1, S = sum_l n_{l-1} n_l."""
    S = sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
    return 6 * C * S - 2 * C * sizes[0] * sizes[1]


def cpu_baseline(npde, wl_small, sets_small, budget_s=14.0):
    """Time the float64 oracle (stencil mode = the reference's algorithm: 6 batched forward passes per Poisson
    residual + reverse mode) on a bounded sample of the same workload on this box's host cores.  torch's intra-op
    thread count is chosen from a short probe (more threads than ~32 slow these small float64 GEMMs down)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import pinn_oracle as po
    import helpers
    ncpu = os.cpu_count() or 1
    prob = helpers.oracle_problem(npde, wl_small.pde_system, wl_small.chains)
    n_int = sets_small[0].shape[1]

    def one():
        t = time.perf_counter()
        po.loss_and_grad(prob, wl_small.theta, sets_small, mode="stencil")
        return time.perf_counter() - t

    best_t, best_n = None, 1
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one()
        dt = min(one(), one())
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
    torch.set_num_threads(best_n)
    t0, reps = time.perf_counter(), 0
    while True:
        one()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 200:
            break
    return {"value": n_int * reps / el, "unit": "interior-point residual+grad evals/s", "cores": best_n, "kind": "port",
            "host_cpus": ncpu,
            "sample": f"{reps} evals of the float64 stencil-mode oracle (torch CPU, {best_n} of {ncpu} hardware threads, best of a "
                      f"8/16/32/64-thread probe) on {n_int} interior + 4x{sets_small[1].shape[1]} boundary points of the same "
                      f"workload; Julia/NeuralPDE.jl itself is not installable here (no network)"}


# HBM-side bytes per launch of the dominant kernel from the PMC passes in profiles/r01_pmc_summary_v11.txt
# (2 x FETCH_SIZE [gfx950 wide-read correction] + WRITE_SIZE, KB -> bytes); only valid for the default workload size.
PMC_TRAFFIC_BYTES = {65536: (2 * 10862.5 + 77600.1) * 1024}    # profiles/r01_pmc_summary_v11.txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events", choices=["dominant", "all", "none"], default="dominant",
                    help="HIP events recorded inside the timed region: around the dominant kernel only (default; each extra "
                         "event pair costs a few us of dispatch gap per step), around every fused kernel, or none")
    ap.add_argument("--event-every", type=int, default=8, help="record the HIP events on every M-th timed step (a start/stop pair plus "
                    "its read-back costs ~20 us of host+dispatch time, 3-10%% of a step; the sampled launches are inside the timed region)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # PINN_BENCH_BACKEND=gloo (+ ranks folded onto the visible devices): functional check of the N > 1 path on a 1-GPU box
    backend = os.environ.get("PINN_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pinn_import
    npde = pinn_import.load()
    from neuralpde_jl_amd import workloads

    wl = workloads.cfg2_poisson2d(points=args.points)
    disc = wl.discretization()
    rep = npde.symbolic_discretize(wl.pde_system, disc)
    eng = rep.engine
    assert eng.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    K, P = eng.K, eng.P
    n_glob = [s.shape[1] for s in sets]
    # shard every term's set into contiguous column blocks (strong scaling)
    if world > 1:
        for k, s in enumerate(sets):
            n = s.shape[1]
            lo, hi = (n * rank) // world, (n * (rank + 1)) // world
            eng.set_points(k, s[:, lo:hi], n_norm=n)
    theta_d = torch.tensor(wl.theta, dtype=torch.float32, device="cuda")
    out_d = torch.zeros(P + K, dtype=torch.float32, device="cuda")
    out_h = torch.zeros(P + K, dtype=torch.float32).pin_memory()
    stream = torch.cuda.current_stream()

    # N = 1: the reduction kernel writes [grad | sums] straight into the pinned (device-mapped, coherent) host buffer — no
    # copy command; N > 1: the all-reduce needs the vector in HBM first, then one D2H copy.
    def step():
        if world > 1:
            eng.loss_grad_device(theta_d.data_ptr(), out_d.data_ptr(), None, stream.cuda_stream)
            dist.all_reduce(out_d)
            out_h.copy_(out_d, non_blocking=True)
        else:
            eng.loss_grad_device(theta_d.data_ptr(), out_h.data_ptr(), None, stream.cuda_stream)
        stream.synchronize()                 # the optimiser needs loss + gradient on the host every iteration

    for _ in range(args.warmup):
        step()
    groups = eng.group_timings()
    dom = int(np.argmax([g["channels"] for g in groups]))      # dominant kernel = the interior (C=5) fused residual kernel
    ev_level, ev_group = {"dominant": 1, "all": 1, "none": 0}[args.events], dom if args.events == "dominant" else -1
    eng.set_timing(ev_level, ev_group)
    step()
    kern_ms = []
    every = max(1, args.event_every)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
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
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    host_path_ms = None
    if world == 1:
        # cross-check of the zero-copy delivery against a plain device-buffer evaluation + copy
        eng.set_timing(0, -1)
        eng.loss_grad_device(theta_d.data_ptr(), out_d.data_ptr(), None, stream.cuda_stream)
        stream.synchronize()
        assert np.array_equal(out_d.cpu().numpy(), out_h.numpy()), "zero-copy host delivery differs from the device buffer"
        # the C-ABI host entry point (theta from host memory, loss + gradient back to host memory: PCIe both ways)
        th = np.ascontiguousarray(wl.theta, dtype=np.float32)
        for _ in range(5):
            eng.loss_grad(th)
        nh = max(10, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(nh):
            eng.loss_grad(th)
        host_path_ms = (time.perf_counter() - t1) / nh * 1e3
    if rank == 0:
        res = out_h.numpy()
        losses = res[P:] / np.array(n_glob)
        groups = eng.group_timings()
        kern_ms = np.array(kern_ms) if kern_ms else np.full((1, len(groups)), np.nan)
        sizes = wl.chains[0].sizes
        dom_ms = float(np.mean(kern_ms[:, dom]))
        # algorithmic flops: the interior residual as written needs C = 5 jet channels (u, u_x, u_y, u_xx, u_yy)
        # = 373,120 flop/point.  The kernel carries u_xx + u_yy as ONE forward-Laplacian channel (C = 4 executed channels,
        # DESIGN.md §2), so it executes fewer flops than the model counts; both figures are reported.
        C_ALG = 5
        flops_dom = algorithmic_flops_per_point(sizes, C_ALG) * groups[dom]["points"]
        flops_exec = algorithmic_flops_per_point(sizes, groups[dom]["channels"]) * groups[dom]["points"]
        achieved = flops_dom / (dom_ms * 1e-3) / 1e12
        all_ms = float(np.mean(kern_ms.sum(axis=1))) if args.events == "all" else None
        flops_all = sum(algorithmic_flops_per_point(sizes, g["channels"]) * g["points"] for g in groups)
        n_int = n_glob[0]
        line = {
            "metric": "collocation-point residual+grad evals/sec, 2D Poisson 4x64 MLP",
            "value": n_int * args.steps / el,
            "unit": "interior-point residual+grad evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.name, "interior_points": n_int, "boundary_terms": K - 1,
                       "boundary_points_per_term": n_glob[1], "theta": P,
                       "parallelism": f"point-shard x{world}" if world > 1 else "single"},
            "point_terms_per_s": sum(n_glob) * args.steps / el,
            "host_entry_ms_per_step": host_path_ms,     # pinn_loss_grad: theta host -> device, results device -> host (PCIe-inclusive)
            "loss_terms": [float(v) for v in losses],
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": PMC_TRAFFIC_BYTES.get(n_int) if world == 1 else None,
                         "traffic_note": "HBM bytes/launch from separate rocprofv3 --pmc passes (profiles/r01_pmc_summary_v11.txt); "
                                         "algorithmic bytes are 8 B/point = 0.5 MB/launch, the rest is the workgroup-private activation-record scratch (L2/Infinity-Cache resident, 16 MB footprint) and the gradient slabs",
                         "kernel": "k_wave2<Spec2<64,3,2,F=xy,LAP=xy>,FUSED> (interior residual+grad, neuron-split workgroups, C=4 executed jet channels)",
                         "kernel_ms": dom_ms, "points_per_launch": groups[dom]["points"],
                         "flops_per_point": algorithmic_flops_per_point(sizes, C_ALG),
                         "executed_channels": groups[dom]["channels"],
                         "executed_flops_per_point": algorithmic_flops_per_point(sizes, groups[dom]["channels"]),
                         "frac_executed": flops_exec / (dom_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "all_fused_kernels_ms": all_ms,
                         "all_fused_kernels_tflops": flops_all / (all_ms * 1e-3) / 1e12 if all_ms else None,
                         "events": f"{args.events} kernel(s), every {every} step(s) of the timed region: {len(kern_ms)} launches averaged"},
        }
        if world == 1 and not args.no_cpu_baseline:
            wls = workloads.cfg2_poisson2d(points=4096)
            reps = npde.symbolic_discretize(wls.pde_system, wls.discretization())
            line["cpu_baseline"] = cpu_baseline(npde, wls, reps.pde_train_sets + reps.bcs_train_sets)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
