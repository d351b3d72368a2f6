"""Hardware mirror of the CPU parity suite: every test of tests/test_emu_parity.py, tests/test_adaptive_losses.py,
tests/test_reference_examples.py, tests/test_jit.py (kernels specialised with hipcc on the box), tests/test_sexpr_frontend.py, tests/test_dgm.py, tests/test_f64_mode.py (the float64 evaluation mode) and tests/test_train_kernel.py (the persistent training kernel against the stand-alone loop, bit for bit) is re-run with the PRODUCT library (libpinn_hip.so on a gfx950 device) instead of the g++ lock-step
emulation — same statements, same float64 oracle, same 1e-5 tolerance.  This is where the per-term gradients (pinn_term_grads), the
BPINN physics log-likelihood, the resident-theta Adam loop (against a host Adam), the device samplers and the adaptive-weight rules
are checked against the oracle ON THE GPU ("Next round" item 1b)."""
import inspect

import pytest

import test_adaptive_losses as ta
import test_dgm as td
import test_emu_parity as tp
import test_f64_mode as tf
import test_jit as tj
import test_reference_examples as tr
import test_sexpr_frontend as ts
import test_train_kernel as tk

pytestmark = pytest.mark.gpu


def _cases():
    out = []
    for mod in (tp, ta, tr, tj, ts, td, tf, tk):
        for name, fn in sorted(vars(mod).items()):
            if not (name.startswith("test_") and inspect.isfunction(fn) and fn.__module__ == mod.__name__):
                continue
            marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
            if not marks:
                out.append(pytest.param(mod, name, {}, id=f"{mod.__name__}.{name}"))
                continue
            assert len(marks) == 1, "one parametrize mark per mirrored test"
            argnames = [a.strip() for a in marks[0].args[0].split(",")]
            for vals in marks[0].args[1]:
                vals = vals if isinstance(vals, (tuple, list)) else (vals,)
                kw = dict(zip(argnames, vals))
                out.append(pytest.param(mod, name, kw, id=f"{mod.__name__}.{name}[{'-'.join(str(v) for v in vals)}]"))
    return out


@pytest.mark.parametrize("mod,name,kw", _cases())
def test_on_hardware(npde, hip_lib, monkeypatch, tmp_path, mod, name, kw):
    monkeypatch.setattr(tp, "EXPECTED_BACKEND", "hip")
    assert npde._lib.default_library().backend == "hip", "the mirror must run on the product library"
    fn = getattr(mod, name)
    args = {}
    for p in inspect.signature(fn).parameters:
        if p == "npde":
            args[p] = npde
        elif p in ("use_emu", "emu_lib"):
            args[p] = None                      # the fixture only switches libraries; the default library is the HIP build here
        elif p == "monkeypatch":
            args[p] = monkeypatch
        elif p == "tmp_path":
            args[p] = tmp_path
        else:
            args[p] = kw[p]
    fn(**args)
