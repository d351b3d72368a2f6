"""The engine-owned data-parallel collective (include/pinn_hip.h: pinn_comm_*, pinn_loss_grad_sharded*).
CPU: the single-process form over 2 and 3 emulated devices equals the single-handle evaluation (shards with n_norm = global N).
GPU: the same entry points through real RCCL on the visible device(s) — a 1-rank communicator on a 1-GPU box."""
import numpy as np
import pytest


def _engines(npde, wl, ndev, devices=None):
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    sets = rep.pde_train_sets + rep.bcs_train_sets
    desc = rep.engine.descriptor
    engs = [npde.Engine(desc, device=(devices[g] if devices else g)) for g in range(ndev)]
    for g, e in enumerate(engs):
        for k, s in enumerate(sets):
            n = s.shape[1]
            lo, hi = (n * g) // ndev, (n * (g + 1)) // ndev
            e.set_points(k, s[:, lo:hi], n_norm=n)
    return rep, engs


@pytest.mark.parametrize("ndev", [2, 3])
def test_single_process_sharded_equals_single_handle(npde, use_emu, ndev):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep, engs = _engines(npde, wl, ndev)
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0], dtype=np.float32)
    L0, G0 = rep.engine.loss_grad(wl.theta, w)
    with pytest.raises(Exception, match="communicator"):
        npde.loss_grad_sharded(engs, wl.theta, w)                     # not initialised yet
    npde.comm_init_all(engs)
    assert [e.comm_size() for e in engs] == [ndev] * ndev
    L, G = npde.loss_grad_sharded(engs, wl.theta, w)
    # the per-term sums of squares are accumulated in double per lane and cross the ranks as doubles: a sharded evaluation returns
    # the single-device losses to double rounding, not merely to float accuracy
    np.testing.assert_allclose(L, L0, rtol=1e-12)
    assert np.linalg.norm(G - G0) / np.linalg.norm(G0) < 1e-6
    L2, G2 = npde.loss_grad_sharded(engs, wl.theta, w)
    assert np.array_equal(L, L2) and np.array_equal(G, G2)             # fixed reduction order
    with pytest.raises(Exception, match="already belongs"):
        npde.comm_init_all(engs)
    with pytest.raises(Exception, match="rank order|communicator"):
        npde.loss_grad_sharded(engs[::-1], wl.theta, w)
    for e in engs:
        e.comm_destroy()
    assert engs[0].comm_size() == 1


def test_init_rank_single_and_errors(npde, use_emu):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=64, bcs_points=16, width=16, hidden=2)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    uid = npde.comm_unique_id()
    assert len(uid) == 128
    with pytest.raises(Exception, match="rank out of range"):
        eng.comm_init_rank(2, 2, uid)
    eng.comm_init_rank(1, 0, uid)
    out = np.zeros(eng.P + eng.K, dtype=np.float32)
    th = np.ascontiguousarray(wl.theta, dtype=np.float32)
    eng.loss_grad_sharded_device(th.ctypes.data, out.ctypes.data, None, 0)         # emulation: "device" pointers are host pointers
    L0, G0 = eng.loss_grad(wl.theta)
    n = np.array([s.shape[1] for s in rep.pde_train_sets + rep.bcs_train_sets])
    np.testing.assert_allclose(out[eng.P:] / n, L0, rtol=1e-6)
    np.testing.assert_array_equal(out[:eng.P], G0)


@pytest.mark.gpu
def test_rccl_communicators_on_the_visible_devices(npde, hip_lib):
    """real RCCL: ncclCommInitAll over every visible device (one on the test boxes) + the grouped all-reduce, and the
    one-process-per-GPU form (ncclGetUniqueId / ncclCommInitRank, 1 rank) with the all-reduce on a caller stream"""
    import torch
    from neuralpde_jl_amd import workloads
    ndev = torch.cuda.device_count()
    wl = workloads.cfg2_poisson2d(points=4096, bcs_points=1024)
    rep, engs = _engines(npde, wl, ndev)
    assert engs[0].L.backend == "hip"
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0], dtype=np.float32)
    L0, G0 = rep.engine.loss_grad(wl.theta, w)
    npde.comm_init_all(engs)
    L, G = npde.loss_grad_sharded(engs, wl.theta, w)
    np.testing.assert_allclose(L, L0, rtol=1e-6)
    assert np.linalg.norm(G - G0) / np.linalg.norm(G0) < 1e-6
    for e in engs:
        e.comm_destroy()
    eng = rep.engine
    eng.comm_init_rank(1, 0, npde.comm_unique_id())
    th = torch.tensor(wl.theta, dtype=torch.float32, device="cuda")
    out = torch.zeros(eng.P + eng.K, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    eng.loss_grad_sharded_device(th.data_ptr(), out.data_ptr(), w, st.cuda_stream)
    st.synchronize()
    n = np.array([s.shape[1] for s in rep.pde_train_sets + rep.bcs_train_sets])
    np.testing.assert_allclose(out[eng.P:].cpu().numpy() / n, L0, rtol=1e-6)
    np.testing.assert_array_equal(out[:eng.P].cpu().numpy(), G0)
    eng.comm_destroy()
