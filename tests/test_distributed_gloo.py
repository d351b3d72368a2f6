"""Multi-GPU path on CPU: world_size-2 gloo processes, each with its own engine handle on one contiguous shard of every
collocation set (n_norm = global count), one all-reduce of [gradient | per-term squared-residual sums] — exactly what
bench.py does over RCCL.  The per-rank compute runs through the emulation build of the kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pinn_import
    m = pinn_import.load()
    from neuralpde_jl_amd import workloads
    m._lib.set_library(m.Library(os.path.join(ROOT, "tests", "emu", "libpinn_emu.so")))
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0], dtype=np.float32)
    full_losses, full_grad = eng.loss_grad(wl.theta, w)
    for k, s in enumerate(sets):
        n = s.shape[1]
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        eng.set_points(k, s[:, lo:hi], n_norm=n)
    theta = torch.tensor(wl.theta, dtype=torch.float32)
    out = torch.zeros(eng.P + eng.K, dtype=torch.float32)
    eng.loss_grad_device(theta.data_ptr(), out.data_ptr(), w, 0)       # "device" pointers are host pointers in the emulation
    dist.all_reduce(out)
    grad = out[: eng.P].numpy()
    losses = out[eng.P:].numpy() / np.array([s.shape[1] for s in sets])
    if rank == 0:
        q.put((np.max(np.abs(losses - full_losses) / full_losses), np.linalg.norm(grad - full_grad) / np.linalg.norm(full_grad)))
    dist.destroy_process_group()


def _adam_worker(rank, world, port, q):
    """the resident Adam loop over a one-process-per-GPU communicator whose transport is torch.distributed (gloo) through
    pinn_comm_init_custom: per iteration evaluate the local shards -> the engine calls back for ONE all-reduce -> the same fused update
    on every rank; no host-side optimiser, theta never leaves the "device" """
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pinn_import
    m = pinn_import.load()
    from neuralpde_jl_amd import workloads
    m._lib.set_library(m.Library(os.path.join(ROOT, "tests", "emu", "libpinn_emu.so")))
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0], dtype=np.float32)
    th_single, hist_single = eng.adam(wl.theta, 6, 1e-2, w)              # the whole sets on one handle
    for k, s in enumerate(sets):
        n = s.shape[1]
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        eng.set_points(k, s[:, lo:hi], n_norm=n)

    def allreduce(buf, count, dtype, stream):
        ty, tt = (ctypes.c_float, torch.float32) if dtype == 0 else (ctypes.c_double, torch.float64)
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ty)), shape=(count,))
        t = torch.from_numpy(a)                                          # shares the engine's buffer: reduced in place
        dist.all_reduce(t)
        return 0

    eng.comm_init_custom(world, rank, allreduce)
    th, hist = eng.adam(wl.theta, 6, 1e-2, w)
    gathered = [None] * world
    dist.all_gather_object(gathered, th)
    if rank == 0:
        same = all(np.array_equal(g, gathered[0]) for g in gathered)
        q.put((same, float(np.max(np.abs(th - th_single))), float(np.max(np.abs(hist - hist_single) / hist_single))))
    eng.comm_destroy()
    dist.destroy_process_group()


def test_two_rank_sharded_resident_adam(emu_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_adam_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, dth, dh = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same                                         # every rank holds the identical theta after the loop
    assert dth < 2e-5 and dh < 1e-5, (dth, dh)          # = the single-handle loop on the whole sets to float accuracy


def test_two_rank_sharded_equals_single(emu_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    le, ge = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert le < 1e-6 and ge < 1e-6, (le, ge)


def _f64_worker(rank, world, port, q):
    """r06: the same two ranks in the FLOAT64 evaluation mode — pinn_loss_grad_sharded_device_f64 and the resident double Adam loop, the
    transport (gloo) called once per evaluation with [P + K] doubles"""
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pinn_import
    m = pinn_import.load()
    from neuralpde_jl_amd import workloads
    m._lib.set_library(m.Library(os.path.join(ROOT, "tests", "emu", "libpinn_emu.so")))
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization(precision="f64"))
    eng = rep.engine
    sets = [np.asarray(s, dtype=np.float64) for s in rep.pde_train_sets + rep.bcs_train_sets]
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0])
    th0 = np.asarray(wl.theta, dtype=np.float64) + 1e-9
    full_losses, full_grad = eng.loss_grad_f64(th0, w)
    th_single, hist_single = eng.adam_f64(th0, 6, 1e-2, w)
    for k, s in enumerate(sets):
        n = s.shape[1]
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        eng.set_points_f64(k, s[:, lo:hi], n_norm=n)
    dtypes = []

    def allreduce(buf, count, dtype, stream):
        dtypes.append(dtype)
        ty = ctypes.c_float if dtype == 0 else ctypes.c_double
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ty)), shape=(count,))
        dist.all_reduce(torch.from_numpy(a))
        return 0

    eng.comm_init_custom(world, rank, allreduce)
    out = np.zeros(eng.P + eng.K)
    eng.loss_grad_sharded_device_f64(th0.ctypes.data, out.ctypes.data, w, 0)
    losses = out[eng.P:] / np.array([s.shape[1] for s in sets])
    le = float(np.max(np.abs(losses - full_losses) / full_losses))
    ge = float(np.linalg.norm(out[:eng.P] - full_grad) / np.linalg.norm(full_grad))
    th, hist = eng.adam_f64(th0, 6, 1e-2, w)
    gathered = [None] * world
    dist.all_gather_object(gathered, th)
    if rank == 0:
        same = all(np.array_equal(g, gathered[0]) for g in gathered)
        q.put((same, le, ge, float(np.max(np.abs(th - th_single))), float(np.max(np.abs(hist - hist_single) / hist_single)), set(dtypes)))
    eng.comm_destroy()
    dist.destroy_process_group()


def test_two_rank_float64_mode(emu_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_f64_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, le, ge, dth, dh, dtypes = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same and dtypes == {1}                       # identical theta on both ranks; only doubles crossed the transport
    assert le < 1e-13 and ge < 1e-13, (le, ge)          # = the single-handle float64 evaluation to double rounding
    assert dth < 2e-9 and dh < 1e-7, (dth, dh)
