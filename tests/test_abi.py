"""The C-ABI library loads on a GPU-less box and exports every symbol include/pinn_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "libpinn_hip.so")
HDR = os.path.join(ROOT, "include", "pinn_hip.h")


def declared_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pinn_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_python_binding_agree(npde):
    assert declared_symbols() == sorted(npde._lib.SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    subprocess.run(["make", "-C", os.path.dirname(LIB), "-j8", "all"], check=True, capture_output=True)      # no-op when up to date
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), f"libpinn_hip.so does not export {s}"
    lib.pinn_backend.restype = ctypes.c_char_p
    assert lib.pinn_backend() == b"hip"
    assert lib.pinn_abi_version() == 1


def test_no_cpu_fallback_without_gpu(npde):
    """On a box without a GPU the product library must refuse to create an engine (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    lib = npde.Library(LIB)
    h = ctypes.c_void_p()
    rc = lib.lib.pinn_create(b"pinnir 1\nntheta 1\n", ctypes.byref(h))
    assert rc != 0 and "no HIP device" in lib.last_error()


def test_hip_objects_contain_gfx950_mfma_code():
    """The shipped kernels are gfx950 code objects using the fp32 MFMA instruction."""
    data = open(LIB, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data          # offload bundle entry for gfx950
    assert b"k_wave" in data                             # the fused residual kernel symbols


def _build_c_client(libdir, libname, out):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_client.c"), "-o", out,
                        "-L" + libdir, "-l:" + libname, "-lm", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return subprocess.run([out], capture_output=True, text=True, timeout=600)


def _u_mid_ok(out):
    m = re.search(r"u\(0\.5\) = ([+-][0-9.]+)", out)              # the trained trial function at the midpoint: sin(pi / 2) = 1
    return m is not None and abs(float(m.group(1)) - 1.0) < 0.02


def test_plain_c_client_of_the_abi(emu_lib, tmp_path):
    """include/pinn_hip.h is a C header (gcc -std=c99 -Werror) and the ABI is usable without any host framework: examples/c_abi_client.c
    (descriptor text in, point sets in, resident Adam, trial function out) compiled by gcc and run against the emulation build."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _build_c_client(os.path.join(root, "tests", "emu"), "libpinn_emu.so", str(tmp_path / "c_abi_client"))
    assert r.returncode == 0 and "backend: emu" in r.stdout and _u_mid_ok(r.stdout), r.stdout + r.stderr


@pytest.mark.gpu
def test_plain_c_client_of_the_abi_gpu(hip_lib, tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _build_c_client(os.path.join(root, "neuralpde.jl_amd", "csrc"), "libpinn_hip.so", str(tmp_path / "c_abi_client"))
    assert r.returncode == 0 and "backend: hip" in r.stdout and _u_mid_ok(r.stdout), r.stdout + r.stderr
