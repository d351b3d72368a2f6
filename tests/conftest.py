import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; runs the product HIP library")


@pytest.fixture(scope="session")
def npde():
    import pinn_import
    return pinn_import.load()


@pytest.fixture(scope="session")
def emu_lib(npde):
    """tests/emu/libpinn_emu.so: the kernel sources compiled by g++ as a 64-lane lock-step emulation.
    TEST INFRASTRUCTURE: lets the CPU suite exercise the real kernel/host code without a GPU.  The product
    never loads it."""
    path = os.path.join(ROOT, "tests", "emu", "libpinn_emu.so")
    csrc = os.path.join(ROOT, "neuralpde.jl_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "emu", "-j8"], check=True, capture_output=True)
    return npde.Library(path)


@pytest.fixture()
def use_emu(npde, emu_lib):
    npde._lib.set_library(emu_lib)
    yield emu_lib
    npde._lib.set_library(None)


@pytest.fixture(scope="session")
def hip_lib(npde):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return npde.Library()
