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


def _host_optimiser_loop(npde, engs, theta, w, nsteps, lr):
    """the loop the sharded resident loop replaces: per iteration theta from the device, pinn_loss_grad_sharded (host pointers in and
    out), the update applied from the host vector (pinn_adam_apply) — same kernels, same all-reduced bits, host round trip every step"""
    e0 = engs[0]
    e0.adam_init(theta)
    n_norm = np.array([1.0] * e0.K)
    hist = []
    for _ in range(nsteps):
        th = e0.adam_get()
        L, G = npde.loss_grad_sharded(engs, th, w)
        hist.append(e0.adam_apply(np.concatenate([G, np.zeros(e0.K, dtype=np.float32)]), lr, w))
        hist[-1] = float(np.dot(np.asarray(w, dtype=np.float64), L))
    return e0.adam_get(), np.array(hist)


@pytest.mark.parametrize("ndev", [2, 3])
def test_sharded_resident_adam_equals_host_optimiser_loop(npde, use_emu, ndev):
    """pinn_adam_steps_sharded over the handles of one communicator: evaluate -> in-stream all-reduce -> fused Adam + weight-image scatter
    on every device, no host synchronisation per iteration — bit-equal to the host-optimiser loop over the same communicator, and equal
    to the single-device resident loop on the union of the shards to float accuracy."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep, engs = _engines(npde, wl, ndev)
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0], dtype=np.float32)
    npde.comm_init_all(engs)
    with pytest.raises(Exception, match="pinn_adam_init"):
        npde.adam_steps_sharded(engs, 3, 1e-2, w)
    with pytest.raises(Exception, match="pinn_adam_steps_sharded"):
        engs[0].adam(wl.theta, 2, 1e-2, w)                             # a single-process communicator's handle: the collective needs all of them
    for e in engs:
        e.adam_init(wl.theta)
    hist = npde.adam_steps_sharded(engs, 6, 1e-2, w)
    thetas = [e.adam_get() for e in engs]
    for t in thetas[1:]:
        assert np.array_equal(t, thetas[0])                            # every rank applied the identical update
    # a second call continues the same optimiser state
    hist2 = npde.adam_steps_sharded(engs, 2, 1e-2, w)
    theta_res = engs[0].adam_get()
    th_host, hist_host = _host_optimiser_loop(npde, engs, wl.theta, w, 8, 1e-2)
    assert np.array_equal(theta_res, th_host)                          # bit-equal: same kernels, same reduction order, same update arithmetic
    np.testing.assert_allclose(np.concatenate([hist, hist2]), hist_host, rtol=1e-6)
    th_single, hist_single = rep.engine.adam(wl.theta, 8, 1e-2, w)
    np.testing.assert_allclose(theta_res, th_single, rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([hist, hist2]), hist_single, rtol=1e-5)
    for e in engs:
        e.comm_destroy()


def test_custom_transport_and_rank_specific_samplers(npde, use_emu):
    """pinn_comm_init_custom: the caller's all-reduce carries the one-process-per-GPU communicator (here: two handles of one process whose
    callbacks add a stored partner vector — the arithmetic of a 2-rank sum).  Device samplers of a communicator's ranks draw different
    points; an un-randomised Sobol design cannot be sharded."""
    import ctypes
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=64, bcs_points=16, width=16, hidden=2)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    calls = []

    def allreduce(buf, count, dtype, stream):
        ty = ctypes.c_float if dtype == 0 else ctypes.c_double
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ty)), shape=(count,))
        a *= 2                                                           # "the other rank holds the same shard": sum = 2 x mine
        calls.append((count, dtype))
        return 0

    L0, G0 = eng.loss_grad(wl.theta)
    eng.comm_init_custom(2, 1, allreduce)
    assert eng.comm_size() == 2
    out = np.zeros(eng.P + eng.K, dtype=np.float32)
    th = np.ascontiguousarray(wl.theta, dtype=np.float32)
    eng.loss_grad_sharded_device(th.ctypes.data, out.ctypes.data, None, 0)
    assert calls == [(eng.P + eng.K, 0), (eng.K, 1)]
    np.testing.assert_array_equal(out[:eng.P], 2 * G0)
    # resident loop over the custom transport: equals plain Adam on the doubled gradient, i.e. (Adam is scale-free up to eps) on G0
    t_c, h_c = eng.adam(wl.theta, 4, 1e-2)
    assert len(calls) == 2 + 2 * 4
    eng.comm_destroy()
    t_p, h_p = eng.adam(wl.theta, 4, 1e-2)
    np.testing.assert_allclose(t_c, t_p, rtol=0, atol=1e-5)
    np.testing.assert_allclose(h_c, 2 * h_p, rtol=1e-4)
    # samplers: rank-specific draws inside a communicator
    lb, ub = np.zeros(2, dtype=np.float32), np.ones(2, dtype=np.float32)
    eng.set_sampler(0, lb, ub, 64, seed=7, kind=1)
    eng.adam(wl.theta, 1, 1e-3)
    p_plain = eng.get_points(0, 2, 64).copy()
    eng.set_sampler(0, lb, ub, 64, seed=7, kind=1)
    eng.comm_init_custom(2, 1, allreduce)
    eng.adam(wl.theta, 1, 1e-3)
    p_rank1 = eng.get_points(0, 2, 64).copy()
    assert not np.array_equal(p_plain, p_rank1)
    eng.set_sampler(0, lb, ub, 64, seed=0, kind=3)
    with pytest.raises(Exception, match="Sobol"):
        eng.adam(wl.theta, 1, 1e-3)
    eng.comm_destroy()


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
    # the resident loop over the real (1-rank) RCCL communicator: in-stream ncclAllReduce between the evaluation and the fused update
    t_c, h_c = eng.adam(wl.theta, 5, 1e-3, w)
    eng.comm_destroy()
    t_p, h_p = eng.adam(wl.theta, 5, 1e-3, w)
    assert np.array_equal(t_c, t_p)
    np.testing.assert_allclose(h_c, h_p, rtol=1e-6)
    # ... and the single-process form over every visible device
    rep2, engs2 = _engines(npde, wl, ndev)
    npde.comm_init_all(engs2)
    for e in engs2:
        e.adam_init(wl.theta)
    h_s = npde.adam_steps_sharded(engs2, 5, 1e-3, w)
    np.testing.assert_allclose(h_s, h_p, rtol=1e-5)
    np.testing.assert_allclose(engs2[0].adam_get(), t_p, rtol=0, atol=2e-5)      # (another handle: Adam amplifies last-bit gradient differences of near-zero entries)
    for e in engs2:
        e.comm_destroy()


def _engines_f64(npde, wl, ndev, devices=None):
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization(precision="f64"))
    sets = rep.pde_train_sets + rep.bcs_train_sets
    desc = rep.engine.descriptor
    engs = [npde.Engine(desc, device=(devices[g] if devices else g)) for g in range(ndev)]
    for g, e in enumerate(engs):
        e.set_option("precision", "f64")
        for k, s in enumerate(sets):
            n = s.shape[1]
            lo, hi = (n * g) // ndev, (n * (g + 1)) // ndev
            e.set_points_f64(k, np.asarray(s, dtype=np.float64)[:, lo:hi], n_norm=n)
    return rep, engs


@pytest.mark.parametrize("ndev", [2, 3])
def test_float64_mode_sharded_equals_single_handle(npde, use_emu, ndev):
    """r06: the communicator paths in the float64 evaluation mode — pinn_loss_grad_sharded_f64 (one all-reduce of [P + K] DOUBLES) returns the
    single-handle float64 losses and gradient to double rounding; pinn_adam_steps_sharded over float64-mode handles equals the single-handle
    resident double loop on the union of the shards"""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=100, bcs_points=37, width=16, hidden=2)
    rep, engs = _engines_f64(npde, wl, ndev)
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0])
    th0 = np.asarray(wl.theta, dtype=np.float64) + 1e-9                # (not representable in float: a float leg anywhere would show)
    L0, G0 = rep.engine.loss_grad_f64(th0, w)
    assert rep.engine.get_option("precision") == "f64"
    npde.comm_init_all(engs)
    L, G = npde.loss_grad_sharded_f64(engs, th0, w)
    np.testing.assert_allclose(L, L0, rtol=1e-13)
    assert np.linalg.norm(G - G0) / np.linalg.norm(G0) < 1e-13
    L2, G2 = npde.loss_grad_sharded_f64(engs, th0, w)
    assert np.array_equal(L, L2) and np.array_equal(G, G2)
    # a float-mode handle in the list is refused
    engs[1].set_option("precision", "f32")
    with pytest.raises(Exception, match="float64 evaluation mode"):
        npde.loss_grad_sharded_f64(engs, th0, w)
    engs[1].set_option("precision", "f64")
    # the resident loop
    for e in engs:
        e.adam_init_f64(th0)
    hist = npde.adam_steps_sharded(engs, 6, 1e-2, w)
    thetas = [e.adam_get_f64() for e in engs]
    for t in thetas[1:]:
        assert np.array_equal(t, thetas[0])
    th_single, hist_single = rep.engine.adam_f64(th0, 6, 1e-2, w)
    np.testing.assert_allclose(thetas[0], th_single, rtol=0, atol=2e-9)      # (Adam amplifies last-bit gradient differences of near-zero entries)
    np.testing.assert_allclose(hist, hist_single, rtol=1e-7)
    assert np.abs(thetas[0] - thetas[0].astype(np.float32)).max() > 1e-10   # the iterate stayed in double
    for e in engs:
        e.comm_destroy()


def test_float64_mode_custom_transport(npde, use_emu):
    """one-process-per-GPU communicator in float64 mode: pinn_loss_grad_sharded_device_f64 hands the caller's transport ONE buffer of
    [P + K] doubles (dtype 1); pinn_adam_steps over it runs the double loop with that all-reduce inside every iteration"""
    import ctypes
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=64, bcs_points=16, width=16, hidden=2)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization(precision="f64"))
    eng = rep.engine
    calls = []

    def allreduce(buf, count, dtype, stream):
        ty = ctypes.c_float if dtype == 0 else ctypes.c_double
        a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ty)), shape=(count,))
        a *= 2
        calls.append((count, dtype))
        return 0

    th0 = np.asarray(wl.theta, dtype=np.float64) + 1e-9
    out = np.zeros(eng.P + eng.K)
    with pytest.raises(Exception, match="no communicator"):
        eng.loss_grad_sharded_device_f64(th0.ctypes.data, out.ctypes.data, None, 0)
    L0, G0 = eng.loss_grad_f64(th0)
    eng.comm_init_custom(2, 1, allreduce)
    eng.loss_grad_sharded_device_f64(th0.ctypes.data, out.ctypes.data, None, 0)
    assert calls == [(eng.P + eng.K, 1)]
    np.testing.assert_array_equal(out[:eng.P], 2 * G0)
    n = np.array([s.shape[1] for s in rep.pde_train_sets + rep.bcs_train_sets])
    np.testing.assert_allclose(out[eng.P:] / n, 2 * L0, rtol=1e-14)
    t_c, h_c = eng.adam_f64(th0, 4, 1e-2)
    assert calls[1:] == [(eng.P + eng.K, 1)] * 4
    eng.comm_destroy()
    t_p, h_p = eng.adam_f64(th0, 4, 1e-2)
    np.testing.assert_allclose(t_c, t_p, rtol=0, atol=1e-5)             # Adam on 2g = Adam on g up to eps
    np.testing.assert_allclose(h_c, 2 * h_p, rtol=1e-4)
    # a float-mode handle is refused by the double entry point
    eng.set_option("precision", "f32")
    eng.comm_init_custom(2, 1, allreduce)
    with pytest.raises(Exception, match="float64 evaluation mode"):
        eng.loss_grad_sharded_device_f64(th0.ctypes.data, out.ctypes.data, None, 0)
    eng.comm_destroy()


@pytest.mark.gpu
def test_rccl_float64_mode_on_the_visible_devices(npde, hip_lib):
    """real RCCL with ncclDouble: the float64-mode communicator paths on the visible device(s) — single-process form, one-process-per-GPU
    form on a caller stream, and the resident double Adam loop with the in-stream all-reduce"""
    import torch
    from neuralpde_jl_amd import workloads
    ndev = torch.cuda.device_count()
    wl = workloads.cfg2_poisson2d(points=4096, bcs_points=1024)
    rep, engs = _engines_f64(npde, wl, ndev)
    assert engs[0].L.backend == "hip"
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0])
    th0 = np.asarray(wl.theta, dtype=np.float64) + 1e-9
    L0, G0 = rep.engine.loss_grad_f64(th0, w)
    assert rep.engine.get_option("f64_path") == "mfma"
    npde.comm_init_all(engs)
    L, G = npde.loss_grad_sharded_f64(engs, th0, w)
    np.testing.assert_allclose(L, L0, rtol=1e-13)
    assert np.linalg.norm(G - G0) / np.linalg.norm(G0) < 1e-13
    for e in engs:
        e.adam_init_f64(th0)
    h_s = npde.adam_steps_sharded(engs, 5, 1e-3, w)
    t_s = engs[0].adam_get_f64()
    for e in engs:
        e.comm_destroy()
    eng = rep.engine
    t_p, h_p = eng.adam_f64(th0, 5, 1e-3, w)
    np.testing.assert_allclose(h_s, h_p, rtol=1e-9)
    np.testing.assert_allclose(t_s, t_p, rtol=0, atol=1e-9)
    eng.comm_init_rank(1, 0, npde.comm_unique_id())
    th = torch.tensor(th0, dtype=torch.float64, device="cuda")
    out = torch.zeros(eng.P + eng.K, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream()
    eng.loss_grad_sharded_device_f64(th.data_ptr(), out.data_ptr(), w, st.cuda_stream)
    st.synchronize()
    n = np.array([s.shape[1] for s in rep.pde_train_sets + rep.bcs_train_sets])
    np.testing.assert_allclose(out[eng.P:].cpu().numpy() / n, L0, rtol=1e-14)
    np.testing.assert_array_equal(out[:eng.P].cpu().numpy(), G0)
    t_c, h_c = eng.adam_f64(th0, 5, 1e-3, w)
    eng.comm_destroy()
    assert np.array_equal(t_c, t_p)
    np.testing.assert_allclose(h_c, h_p, rtol=1e-14)
