"""ctypes binding of the C ABI in include/pinn_hip.h (libpinn_hip.so).

This is the Python mirror of the `ccall` layer the Julia glue uses.  There is no
CPU fallback: if the HIP library is missing this module raises, and if no gfx950 device is
present `pinn_create` fails with the library's own error message.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

# PyTorch ships its own copy of the HIP / HSA runtime (torch/lib/libamdhip64.so).  A process that maps libpinn_hip.so FIRST binds the
# engine to the system runtime (/opt/rocm) and then initialises a SECOND runtime when torch touches the device: the later one finds "no
# ROCm-capable device" (seen on hardware: build() followed by smoke() in one process).  With torch mapped first, the engine's HIP symbols
# resolve to the runtime torch already carries, and the two share one device context — which the zero-copy paths (device pointers and
# streams handed across the ABI) need anyway.  Hosts without torch (the Julia glue, the C client) use the system runtime alone.
try:
    import torch as _torch  # noqa: F401
except ImportError:          # pragma: no cover
    _torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "csrc", "libpinn_hip.so")

# every symbol declared in include/pinn_hip.h
SYMBOLS = [
    "pinn_backend", "pinn_abi_version", "pinn_last_error", "pinn_create", "pinn_destroy", "pinn_num_terms",
    "pinn_num_theta", "pinn_set_points", "pinn_set_points_device", "pinn_loss_grad", "pinn_loss_grad_f64",
    "pinn_term_grads", "pinn_loglik_grad", "pinn_loss_grad_device", "pinn_residual", "pinn_phi", "pinn_derivative", "pinn_last_timing", "pinn_set_timing", "pinn_describe", "pinn_num_groups", "pinn_group_timing",
    "pinn_create_on", "pinn_comm_unique_id", "pinn_comm_init_rank", "pinn_comm_init_all", "pinn_comm_size", "pinn_comm_rank", "pinn_comm_destroy",
    "pinn_loss_grad_sharded_device", "pinn_loss_grad_sharded",
    "pinn_set_sampler", "pinn_set_point_data", "pinn_set_point_weights", "pinn_get_points", "pinn_adam_init", "pinn_adam_steps", "pinn_adam_get", "pinn_lbfgs",
    "pinn_loss_device", "pinn_group_launched_by",
    "pinn_set_option", "pinn_get_option", "pinn_set_points_f64", "pinn_comm_init_custom", "pinn_adam_steps_sharded", "pinn_adam_apply",
    "pinn_adam_init_f64", "pinn_adam_get_f64", "pinn_set_point_data_f64", "pinn_loss_grad_device_f64",
    "pinn_residual_f64", "pinn_phi_f64", "pinn_derivative_f64", "pinn_term_grads_f64", "pinn_loglik_grad_f64",
    "pinn_loss_grad_sharded_device_f64", "pinn_loss_grad_sharded_f64",
]


class EngineError(RuntimeError):
    pass


class Library:
    """A loaded build of the engine ABI."""

    def __init__(self, path: str = DEFAULT_PATH):
        if not os.path.exists(path):
            raise EngineError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "The PINN engine has no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
        L.pinn_backend.restype = C.c_char_p
        L.pinn_abi_version.restype = C.c_int
        L.pinn_last_error.restype = C.c_char_p
        L.pinn_create.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.pinn_create_on.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.pinn_comm_unique_id.argtypes = [vp, C.c_int64]
        L.pinn_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int64]
        L.pinn_comm_init_all.argtypes = [C.POINTER(vp), C.c_int]
        L.pinn_comm_size.argtypes = [vp]
        L.pinn_comm_rank.argtypes = [vp]
        L.pinn_comm_destroy.argtypes = [vp]
        L.pinn_loss_grad_sharded_device.argtypes = [vp, vp, fp, vp, vp]
        L.pinn_loss_grad_sharded.argtypes = [C.POINTER(vp), C.c_int, fp, C.c_int64, fp, dp, fp]
        L.pinn_destroy.argtypes = [vp]
        L.pinn_num_terms.argtypes = [vp]
        L.pinn_num_theta.argtypes = [vp]
        L.pinn_num_theta.restype = C.c_int64
        L.pinn_set_points.argtypes = [vp, C.c_int, fp, C.c_int64, C.c_int64]
        L.pinn_set_points_device.argtypes = [vp, C.c_int, vp, C.c_int64, C.c_int64]
        L.pinn_loss_grad.argtypes = [vp, fp, C.c_int64, fp, dp, fp]
        L.pinn_loss_grad_f64.argtypes = [vp, dp, C.c_int64, dp, dp, dp]
        L.pinn_term_grads.argtypes = [vp, fp, C.c_int64, dp, fp]
        L.pinn_loglik_grad.argtypes = [vp, fp, C.c_int64, dp, dp, fp, dp]
        L.pinn_loss_grad_device.argtypes = [vp, vp, fp, vp, vp]
        L.pinn_loss_device.argtypes = [vp, vp, vp, vp]
        L.pinn_group_launched_by.argtypes = [vp, C.c_int]
        L.pinn_residual.argtypes = [vp, C.c_int, fp, C.c_int64, fp]
        L.pinn_phi.argtypes = [vp, C.c_int, fp, C.c_int64, fp, C.c_int64, fp]
        L.pinn_derivative.argtypes = [vp, C.c_int, fp, C.c_int64, fp, C.c_int64, C.c_int, C.POINTER(C.c_int), fp]
        L.pinn_last_timing.argtypes = [vp, fp, fp]
        L.pinn_set_timing.argtypes = [vp, C.c_int, C.c_int]
        L.pinn_describe.argtypes = [vp, C.c_char_p, C.c_int64]
        L.pinn_num_groups.argtypes = [vp]
        L.pinn_set_sampler.argtypes = [vp, C.c_int, C.c_int, fp, fp, C.c_int64, C.c_uint64]
        L.pinn_get_points.argtypes = [vp, C.c_int, fp, C.c_int64]
        L.pinn_set_point_data.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int64]
        L.pinn_set_point_weights.argtypes = [vp, C.c_int, fp, C.c_int64]
        L.pinn_adam_init.argtypes = [vp, fp, C.c_int64]
        L.pinn_adam_steps.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, fp, dp]
        L.pinn_adam_get.argtypes = [vp, fp, C.c_int64]
        try:                                     # (r05 entry points: A/B variant libraries built before them still load for tools/ab_compare.py)
            L.pinn_adam_init_f64.argtypes = [vp, dp, C.c_int64]
            L.pinn_adam_get_f64.argtypes = [vp, dp, C.c_int64]
            L.pinn_set_point_data_f64.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int64]
            L.pinn_loss_grad_device_f64.argtypes = [vp, vp, fp, vp, vp]
            L.pinn_residual_f64.argtypes = [vp, C.c_int, dp, C.c_int64, dp]
            L.pinn_phi_f64.argtypes = [vp, C.c_int, dp, C.c_int64, dp, C.c_int64, dp]
            L.pinn_derivative_f64.argtypes = [vp, C.c_int, dp, C.c_int64, dp, C.c_int64, C.c_int, C.POINTER(C.c_int), dp]
            L.pinn_term_grads_f64.argtypes = [vp, dp, C.c_int64, dp, dp]
            L.pinn_loglik_grad_f64.argtypes = [vp, dp, C.c_int64, dp, dp, dp, dp]
            L.pinn_loss_grad_sharded_device_f64.argtypes = [vp, vp, fp, vp, vp]
            L.pinn_loss_grad_sharded_f64.argtypes = [C.POINTER(vp), C.c_int, dp, C.c_int64, dp, dp, dp]
        except AttributeError:
            pass
        L.pinn_lbfgs.argtypes = [vp, C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_int, C.c_double, fp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        self.ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int64, C.c_int, vp)
        L.pinn_comm_init_custom.argtypes = [vp, C.c_int, C.c_int, self.ALLREDUCE_FN, vp]
        L.pinn_adam_steps_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, fp, dp]
        L.pinn_adam_apply.argtypes = [vp, fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, fp, dp]
        L.pinn_set_points_f64.argtypes = [vp, C.c_int, dp, C.c_int64, C.c_int64]
        L.pinn_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.pinn_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int64]
        L.pinn_group_timing.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]

    @property
    def backend(self) -> str:
        return self.lib.pinn_backend().decode()

    def last_error(self) -> str:
        return self.lib.pinn_last_error().decode()

    def check(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what}: {self.last_error()}")


_default: Optional[Library] = None


def default_library() -> Library:
    global _default
    if _default is None:
        _default = Library(DEFAULT_PATH)
    return _default


def set_library(lib: Optional[Library]):
    """Replace the process-wide library object.  Used by the CPU test-suite to drive the host logic
    through tests/emu/libpinn_emu.so (same sources, wave emulation); never called by the package."""
    global _default
    _default = lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


COMM_ID_BYTES = 128


def comm_unique_id(lib: Optional[Library] = None) -> bytes:
    """rank 0 of a one-process-per-GPU job: the 128-byte id every rank passes to Engine.comm_init_rank"""
    L = lib or default_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    L.check(L.lib.pinn_comm_unique_id(buf, COMM_ID_BYTES), "pinn_comm_unique_id")
    return buf.raw


def comm_init_all(engines: Sequence["Engine"]):
    """single process, one engine per device (Engine(descriptor, device=g)): form their communicator (ncclCommInitAll)"""
    L = engines[0].L
    hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
    L.check(L.lib.pinn_comm_init_all(hs, len(engines)), "pinn_comm_init_all")


def loss_grad_sharded(engines: Sequence["Engine"], theta, weights=None, want_grad: bool = True):
    """one evaluation over all devices of a comm_init_all communicator: every engine holds its shard of every term's point set
    (set_points(..., n_norm = global N)); returns the GLOBAL per-term losses and gradient (pinn_loss_grad_sharded)"""
    e0 = engines[0]
    L = e0.L
    hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
    th = _f32(theta)
    losses = np.zeros(e0.K, dtype=np.float64)
    grad = np.zeros(e0.P, dtype=np.float32) if want_grad else None
    w = _f32(weights) if weights is not None else None
    L.check(L.lib.pinn_loss_grad_sharded(hs, len(engines), th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
                                         w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                                         losses.ctypes.data_as(C.POINTER(C.c_double)),
                                         grad.ctypes.data_as(C.POINTER(C.c_float)) if grad is not None else None), "pinn_loss_grad_sharded")
    return losses, grad


def loss_grad_sharded_f64(engines: Sequence["Engine"], theta, weights=None, want_grad: bool = True):
    """`pinn_loss_grad_sharded_f64`: the same over handles in float64 mode — theta, losses and gradient in double, one all-reduce of [P + K] doubles"""
    e0 = engines[0]
    L = e0.L
    hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
    th = _f64(theta)
    losses = np.zeros(e0.K, dtype=np.float64)
    grad = np.zeros(e0.P, dtype=np.float64) if want_grad else None
    w = _f64(weights) if weights is not None else None
    L.check(L.lib.pinn_loss_grad_sharded_f64(hs, len(engines), th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
                                             w.ctypes.data_as(C.POINTER(C.c_double)) if w is not None else None,
                                             losses.ctypes.data_as(C.POINTER(C.c_double)),
                                             grad.ctypes.data_as(C.POINTER(C.c_double)) if grad is not None else None), "pinn_loss_grad_sharded_f64")
    return losses, grad


def adam_steps_sharded(engines: Sequence["Engine"], nsteps: int, lr: float, weights=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """`pinn_adam_steps_sharded`: the resident Adam loop over the handles of one `comm_init_all` communicator (every handle
    `adam_init`-ed with the same theta): per iteration evaluate the local shards, ONE all-reduce, the same update on every device.
    Returns the loss history (rank 0's)."""
    L = engines[0].L
    hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
    hist = np.zeros(nsteps, dtype=np.float64)
    w = _f32(weights) if weights is not None else None
    L.check(L.lib.pinn_adam_steps_sharded(hs, len(engines), nsteps, lr, beta1, beta2, eps,
                                          w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                                          hist.ctypes.data_as(C.POINTER(C.c_double))), "pinn_adam_steps_sharded")
    return hist


class Engine:
    """One `pinn_handle`."""

    def __init__(self, descriptor: str, lib: Optional[Library] = None, device: int = -1):
        """device: HIP device index for single-process multi-GPU use (pinn_create_on); -1 = the caller's current device"""
        self.L = lib or default_library()
        self.h = C.c_void_p()
        self.descriptor = descriptor
        self.L.check(self.L.lib.pinn_create_on(descriptor.encode(), device, C.byref(self.h)), "pinn_create")
        self.K = self.L.lib.pinn_num_terms(self.h)
        self.P = int(self.L.lib.pinn_num_theta(self.h))
        # (no HIP events around the kernels of an evaluation unless a caller asks for timings with set_timing: the library's default since
        # r04 — the phase events behind pinn_last_timing cost 25 us per host-entry call, as much as a small problem's kernels)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.lib.pinn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_points(self, term: int, pts, n_norm: int = 0):
        """pts: (d x N) array as the reference holds it; stored point-major (Julia column-major)."""
        pts = np.asarray(pts)
        flat = _f32(pts.T).reshape(-1)     # (N x d) C-order == (d x N) Fortran order
        self.L.check(self.L.lib.pinn_set_points(self.h, term, flat.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[1], n_norm),
                     "pinn_set_points")

    def set_points_f64(self, term: int, pts, n_norm: int = 0):
        """`pinn_set_points_f64`: the set in double for the float64 evaluation mode (the fp32 kernels get its float conversion)"""
        pts = np.asarray(pts, dtype=np.float64)
        flat = np.ascontiguousarray(pts.T).reshape(-1)
        self.L.check(self.L.lib.pinn_set_points_f64(self.h, term, flat.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[1], n_norm), "pinn_set_points_f64")

    def set_points_device(self, term: int, dptr: int, n: int, n_norm: int = 0):
        self.L.check(self.L.lib.pinn_set_points_device(self.h, term, C.c_void_p(dptr), n, n_norm), "pinn_set_points_device")

    def loss_grad(self, theta, weights: Optional[Sequence[float]] = None, want_grad: bool = True):
        """K term losses (+ gradient).  want_grad=False is the LOSS-ONLY evaluation (grad = NULL at the ABI): forward pass + residuals +
        sums of squares, no reverse sweep — the same loss values at about a third of the cost."""
        th = _f32(theta)
        losses = np.zeros(self.K, dtype=np.float64)
        grad = np.zeros(self.P, dtype=np.float32) if want_grad else None
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad(
            self.h, th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
            w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
            losses.ctypes.data_as(C.POINTER(C.c_double)),
            grad.ctypes.data_as(C.POINTER(C.c_float)) if grad is not None else None), "pinn_loss_grad")
        return losses, grad

    def loss_grad_f64(self, theta, weights=None, want_grad: bool = True):
        """theta, weights, losses and gradient in double at the ABI.  With the handle's precision option at "f64" the evaluation itself runs
        in double; otherwise the fp32 kernels run and the results are widened."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
        losses = np.zeros(self.K, dtype=np.float64)
        grad = np.zeros(self.P, dtype=np.float64) if want_grad else None
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64)) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad_f64(
            self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
            w.ctypes.data_as(C.POINTER(C.c_double)) if w is not None else None,
            losses.ctypes.data_as(C.POINTER(C.c_double)),
            grad.ctypes.data_as(C.POINTER(C.c_double)) if grad is not None else None), "pinn_loss_grad_f64")
        return losses, grad

    def term_grads(self, theta):
        th = _f32(theta)
        losses = np.zeros(self.K, dtype=np.float64)
        tg = np.zeros((self.K, self.P), dtype=np.float32)
        self.L.check(self.L.lib.pinn_term_grads(self.h, th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
                                                losses.ctypes.data_as(C.POINTER(C.c_double)),
                                                tg.ctypes.data_as(C.POINTER(C.c_float))), "pinn_term_grads")
        return losses, tg

    def term_grads_f64(self, theta):
        """`pinn_term_grads_f64`: theta and the K x P per-term gradients in double (native in float64 mode)"""
        th = _f64(theta)
        losses = np.zeros(self.K, dtype=np.float64)
        tg = np.zeros((self.K, self.P), dtype=np.float64)
        self.L.check(self.L.lib.pinn_term_grads_f64(self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
                                                    losses.ctypes.data_as(C.POINTER(C.c_double)),
                                                    tg.ctypes.data_as(C.POINTER(C.c_double))), "pinn_term_grads_f64")
        return losses, tg

    def loglik_grad_f64(self, theta, stds, want_grad: bool = True):
        """`pinn_loglik_grad_f64`: theta and d loglik / d theta in double (native in float64 mode)"""
        th = _f64(theta)
        sd = _f64(stds)
        if sd.shape != (self.K,):
            raise ValueError(f"need one std per loss term ({self.K})")
        ll = C.c_double()
        g = np.zeros(self.P, dtype=np.float64) if want_grad else None
        gs = np.zeros(self.K, dtype=np.float64)
        self.L.check(self.L.lib.pinn_loglik_grad_f64(self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size, sd.ctypes.data_as(C.POINTER(C.c_double)),
                                                     C.byref(ll), g.ctypes.data_as(C.POINTER(C.c_double)) if g is not None else None,
                                                     gs.ctypes.data_as(C.POINTER(C.c_double))), "pinn_loglik_grad_f64")
        return ll.value, g, gs

    def loglik_grad(self, theta, stds, want_grad: bool = True):
        """BPINN log-likelihood sum_k logpdf(MvNormal(r_k, std_k^2 I), 0), d/dtheta and d/dstd (pinn_loglik_grad)"""
        th = _f32(theta)
        sd = np.ascontiguousarray(np.asarray(stds, dtype=np.float64))
        if sd.shape != (self.K,):
            raise ValueError(f"need one std per loss term ({self.K})")
        ll = C.c_double()
        g = np.zeros(self.P, dtype=np.float32) if want_grad else None
        gs = np.zeros(self.K, dtype=np.float64)
        self.L.check(self.L.lib.pinn_loglik_grad(self.h, th.ctypes.data_as(C.POINTER(C.c_float)), th.size, sd.ctypes.data_as(C.POINTER(C.c_double)),
                                                 C.byref(ll), g.ctypes.data_as(C.POINTER(C.c_float)) if g is not None else None,
                                                 gs.ctypes.data_as(C.POINTER(C.c_double))), "pinn_loglik_grad")
        return ll.value, g, gs

    def loss_grad_device(self, d_theta: int, d_out: int, weights=None, stream: int = 0):
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad_device(
            self.h, C.c_void_p(d_theta), w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
            C.c_void_p(d_out), C.c_void_p(stream)), "pinn_loss_grad_device")

    def loss_grad_device_f64(self, d_theta: int, d_out: int, weights=None, stream: int = 0):
        """`pinn_loss_grad_device_f64`: a float64-mode handle on DOUBLE device buffers (d_theta: P doubles, d_out: P + K doubles)"""
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad_device_f64(
            self.h, C.c_void_p(d_theta), w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
            C.c_void_p(d_out), C.c_void_p(stream)), "pinn_loss_grad_device_f64")

    def loss_device(self, d_theta: int, d_sums: int, stream: int = 0):
        """loss-only evaluation on device pointers: d_sums = K floats (this shard's sums of squared residuals per term)"""
        self.L.check(self.L.lib.pinn_loss_device(self.h, C.c_void_p(d_theta), C.c_void_p(d_sums), C.c_void_p(stream)), "pinn_loss_device")

    # ---- engine-owned data parallelism (include/pinn_hip.h: pinn_comm_*) ----
    def comm_init_rank(self, nranks: int, rank: int, uid: bytes):
        """one process per GPU: join the communicator identified by `uid` (comm_unique_id() of rank 0, distributed by the host side)"""
        buf = C.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        self.L.check(self.L.lib.pinn_comm_init_rank(self.h, nranks, rank, buf, COMM_ID_BYTES), "pinn_comm_init_rank")

    def comm_size(self) -> int:
        return self.L.lib.pinn_comm_size(self.h)

    def comm_destroy(self):
        self.L.check(self.L.lib.pinn_comm_destroy(self.h), "pinn_comm_destroy")

    def loss_grad_sharded_device_f64(self, d_theta: int, d_out: int, weights=None, stream: int = 0):
        """`pinn_loss_grad_sharded_device_f64`: float64 mode, DOUBLE device buffers, the engine's all-reduce of [P + K] doubles on `stream`"""
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad_sharded_device_f64(
            self.h, C.c_void_p(d_theta), w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
            C.c_void_p(d_out), C.c_void_p(stream)), "pinn_loss_grad_sharded_device_f64")

    def loss_grad_sharded_device(self, d_theta: int, d_out: int, weights=None, stream: int = 0):
        """pinn_loss_grad_device on this rank's shards + the engine's all-reduce of d_out ([P + K] floats, device memory) on `stream`"""
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_loss_grad_sharded_device(
            self.h, C.c_void_p(d_theta), w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
            C.c_void_p(d_out), C.c_void_p(stream)), "pinn_loss_grad_sharded_device")

    def residual(self, term: int, theta, n: int) -> np.ndarray:
        th = _f32(theta)
        r = np.zeros(n, dtype=np.float32)
        self.L.check(self.L.lib.pinn_residual(self.h, term, th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
                                              r.ctypes.data_as(C.POINTER(C.c_float))), "pinn_residual")
        return r

    def residual_f64(self, term: int, theta, n: int) -> np.ndarray:
        """`pinn_residual_f64`: the term's residuals in double (native in float64 mode: equal to a Float64 evaluation to rounding)"""
        th = _f64(theta)
        r = np.zeros(n, dtype=np.float64)
        self.L.check(self.L.lib.pinn_residual_f64(self.h, term, th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
                                                  r.ctypes.data_as(C.POINTER(C.c_double))), "pinn_residual_f64")
        return r

    def phi_f64(self, net: int, theta, pts) -> np.ndarray:
        """`pinn_phi_f64`: the trial function at the columns of pts (d x N), everything in double"""
        return self.derivative_f64(net, theta, pts, ())

    def derivative_f64(self, net: int, theta, pts, axes: Sequence[int]) -> np.ndarray:
        """`pinn_derivative_f64`: d^k phi_net / dx_axes in double (numeric_derivative at the reference's Float64 tolerances,
        test/Forward/forward__derivatives.jl:29-44)"""
        th = _f64(theta)
        pts = np.asarray(pts, dtype=np.float64)
        flat = np.ascontiguousarray(pts.T).reshape(-1)
        out = np.zeros(pts.shape[1], dtype=np.float64)
        ax = (C.c_int * max(len(axes), 1))(*[int(a) for a in axes])
        if len(axes) == 0:
            self.L.check(self.L.lib.pinn_phi_f64(self.h, net, th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
                                                 flat.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[1],
                                                 out.ctypes.data_as(C.POINTER(C.c_double))), "pinn_phi_f64")
        else:
            self.L.check(self.L.lib.pinn_derivative_f64(self.h, net, th.ctypes.data_as(C.POINTER(C.c_double)), th.size,
                                                        flat.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[1], len(axes), ax,
                                                        out.ctypes.data_as(C.POINTER(C.c_double))), "pinn_derivative_f64")
        return out

    def phi(self, net: int, theta, pts) -> np.ndarray:
        th = _f32(theta)
        pts = np.asarray(pts)
        flat = _f32(pts.T).reshape(-1)
        out = np.zeros(pts.shape[1], dtype=np.float32)
        self.L.check(self.L.lib.pinn_phi(self.h, net, th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
                                         flat.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[1],
                                         out.ctypes.data_as(C.POINTER(C.c_float))), "pinn_phi")
        return out

    def derivative(self, net: int, theta, pts, axes: Sequence[int]) -> np.ndarray:
        """d^k phi_net / dx_axes at the columns of pts (d x N): the engine's counterpart of numeric_derivative (src/pinn_types.jl:445-482)"""
        th = _f32(theta)
        pts = np.asarray(pts)
        flat = _f32(pts.T).reshape(-1)
        out = np.zeros(pts.shape[1], dtype=np.float32)
        ax = (C.c_int * max(len(axes), 1))(*[int(a) for a in axes])
        self.L.check(self.L.lib.pinn_derivative(self.h, net, th.ctypes.data_as(C.POINTER(C.c_float)), th.size,
                                                flat.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[1], len(axes), ax,
                                                out.ctypes.data_as(C.POINTER(C.c_float))), "pinn_derivative")
        return out

    def set_timing(self, level: int, group: int = -1):
        self.L.check(self.L.lib.pinn_set_timing(self.h, level, group), "pinn_set_timing")

    def last_timing(self):
        k, t = C.c_float(), C.c_float()
        self.L.check(self.L.lib.pinn_last_timing(self.h, C.byref(k), C.byref(t)), "pinn_last_timing")
        return k.value, t.value

    def set_point_data(self, term: int, data):
        """data: (ndata x N) per-point channels of the term's current point set (descriptor op DATA j)"""
        data = _f32(np.atleast_2d(np.asarray(data)))
        self.L.check(self.L.lib.pinn_set_point_data(self.h, term, data.ctypes.data_as(C.POINTER(C.c_float)), data.shape[0], data.shape[1]),
                     "pinn_set_point_data")

    def set_point_data_f64(self, term: int, data):
        """`pinn_set_point_data_f64`: the observations in double for the float64 evaluation mode (the fp32 kernels get their float conversion)"""
        data = np.ascontiguousarray(np.atleast_2d(np.asarray(data, dtype=np.float64)))
        self.L.check(self.L.lib.pinn_set_point_data_f64(self.h, term, data.ctypes.data_as(C.POINTER(C.c_double)), data.shape[0], data.shape[1]),
                     "pinn_set_point_data_f64")

    def set_point_weights(self, term: int, w):
        """quadrature weights of the term's current point set: loss_k = sum_i w_i r_i^2 (None: back to mean(abs2, r))"""
        if w is None:
            self.L.check(self.L.lib.pinn_set_point_weights(self.h, term, None, 0), "pinn_set_point_weights")
            return
        w = _f32(np.asarray(w).reshape(-1))
        self.L.check(self.L.lib.pinn_set_point_weights(self.h, term, w.ctypes.data_as(C.POINTER(C.c_float)), w.size), "pinn_set_point_weights")

    def get_points(self, term: int, d: int, n: int) -> np.ndarray:
        """the term's current set as a (d x N) array (e.g. what the device sampler drew last)"""
        buf = np.zeros((n, d), dtype=np.float32)
        self.L.check(self.L.lib.pinn_get_points(self.h, term, buf.ctypes.data_as(C.POINTER(C.c_float)), n), "pinn_get_points")
        return buf.T.copy()

    def set_sampler(self, term: int, lb, ub, n: int, seed: int = 0, kind: int = 1):
        lb, ub = _f32(lb), _f32(ub)
        self.L.check(self.L.lib.pinn_set_sampler(self.h, term, kind, lb.ctypes.data_as(C.POINTER(C.c_float)),
                                                 ub.ctypes.data_as(C.POINTER(C.c_float)), n, seed), "pinn_set_sampler")

    def adam(self, theta, nsteps: int, lr: float, weights=None, beta1=0.9, beta2=0.999, eps=1e-8, init=True):
        """`nsteps` Adam iterations with theta resident on the device; returns (theta, loss history).  init=False continues from the
        optimiser state on the device (theta is not read)."""
        th = _f32(theta) if init else None
        if init:
            self.L.check(self.L.lib.pinn_adam_init(self.h, th.ctypes.data_as(C.POINTER(C.c_float)), th.size), "pinn_adam_init")
        hist = np.zeros(nsteps, dtype=np.float64)
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_adam_steps(self.h, nsteps, lr, beta1, beta2, eps,
                                                w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                                                hist.ctypes.data_as(C.POINTER(C.c_double))), "pinn_adam_steps")
        out = np.zeros(self.P, dtype=np.float32)
        self.L.check(self.L.lib.pinn_adam_get(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "pinn_adam_get")
        return out, hist

    def adam_f64(self, theta, nsteps: int, lr: float, weights=None, beta1=0.9, beta2=0.999, eps=1e-8, init=True):
        """the same through the double entry points (pinn_adam_init_f64 / pinn_adam_get_f64): on a handle in float64 mode every kernel of the
        loop — redraw, residual + gradient, Adam — runs in double on the device; returns (theta float64, loss history)"""
        if init:
            th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
            self.L.check(self.L.lib.pinn_adam_init_f64(self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size), "pinn_adam_init_f64")
        hist = np.zeros(nsteps, dtype=np.float64)
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_adam_steps(self.h, nsteps, lr, beta1, beta2, eps,
                                                w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                                                hist.ctypes.data_as(C.POINTER(C.c_double))), "pinn_adam_steps")
        out = np.zeros(self.P, dtype=np.float64)
        self.L.check(self.L.lib.pinn_adam_get_f64(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size), "pinn_adam_get_f64")
        return out, hist

    def lbfgs(self, theta, maxiters: int, weights=None, history: int = 10, gtol: float = 1e-8):
        """`pinn_lbfgs`: L-BFGS on the weighted objective over the installed (fixed) point sets; returns (theta float64, objective after
        every performed iteration)."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64)).copy()
        hist = np.zeros(int(maxiters), dtype=np.float64)
        done = C.c_int(0)
        w = _f32(weights) if weights is not None else None
        self.L.check(self.L.lib.pinn_lbfgs(self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size, int(maxiters), int(history), float(gtol),
                                           w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                                           hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done)), "pinn_lbfgs")
        return th, hist[:done.value]

    def adam_init_f64(self, theta):
        th = _f64(theta)
        self.L.check(self.L.lib.pinn_adam_init_f64(self.h, th.ctypes.data_as(C.POINTER(C.c_double)), th.size), "pinn_adam_init_f64")

    def adam_get_f64(self) -> np.ndarray:
        out = np.zeros(self.P, dtype=np.float64)
        self.L.check(self.L.lib.pinn_adam_get_f64(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size), "pinn_adam_get_f64")
        return out

    def adam_init(self, theta):
        th = _f32(theta)
        self.L.check(self.L.lib.pinn_adam_init(self.h, th.ctypes.data_as(C.POINTER(C.c_float)), th.size), "pinn_adam_init")

    def adam_get(self) -> np.ndarray:
        out = np.zeros(self.P, dtype=np.float32)
        self.L.check(self.L.lib.pinn_adam_get(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "pinn_adam_get")
        return out

    def adam_apply(self, grad_and_sums, lr: float, weights=None, beta1=0.9, beta2=0.999, eps=1e-8) -> float:
        """`pinn_adam_apply`: one Adam update of the resident state from a host vector [gradient (P) | raw per-term sums (K)]"""
        v = _f32(grad_and_sums)
        w = _f32(weights) if weights is not None else None
        loss = C.c_double(0.0)
        self.L.check(self.L.lib.pinn_adam_apply(self.h, v.ctypes.data_as(C.POINTER(C.c_float)), v.size, lr, beta1, beta2, eps,
                                                w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None, C.byref(loss)), "pinn_adam_apply")
        return loss.value

    def comm_init_custom(self, nranks: int, rank: int, allreduce):
        """`pinn_comm_init_custom`: `allreduce(buf_ptr, count, dtype, stream_ptr) -> int` (dtype 0 float / 1 double) becomes the transport of
        this handle's communicator; the ctypes callback object is kept alive by the engine wrapper."""
        def _cb(ctx, buf, count, dtype, stream):
            try:
                return int(allreduce(buf, count, dtype, stream) or 0)
            except Exception:                                           # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._allreduce_cb = self.L.ALLREDUCE_FN(_cb)
        self.L.check(self.L.lib.pinn_comm_init_custom(self.h, nranks, rank, self._allreduce_cb, None), "pinn_comm_init_custom")

    def set_option(self, name: str, value: str):
        """run-time options of the handle (include/pinn_hip.h: pinn_set_option); "gemm" = "split" | "fp32" """
        self.L.check(self.L.lib.pinn_set_option(self.h, name.encode(), value.encode()), "pinn_set_option")

    def get_option(self, name: str) -> str:
        buf = C.create_string_buffer(64)
        self.L.check(self.L.lib.pinn_get_option(self.h, name.encode(), buf, 64), "pinn_get_option")
        return buf.value.decode()

    def group_timings(self):
        out = []
        for g in range(self.L.lib.pinn_num_groups(self.h)):
            ms, pts, ch, tiles = C.c_float(), C.c_int64(), C.c_int(), C.c_int()
            self.L.check(self.L.lib.pinn_group_timing(self.h, g, C.byref(ms), C.byref(pts), C.byref(ch), C.byref(tiles)),
                         "pinn_group_timing")
            out.append(dict(group=g, ms=ms.value, points=pts.value, channels=ch.value, tiles=tiles.value,
                            launched_by=self.L.lib.pinn_group_launched_by(self.h, g)))
        return out

    def describe(self) -> str:
        buf = C.create_string_buffer(4096)
        self.L.check(self.L.lib.pinn_describe(self.h, buf, 4096), "pinn_describe")
        return buf.value.decode()
