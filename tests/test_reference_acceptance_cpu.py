"""The quick cases of tests/test_gpu_reference_acceptance.py (the reference's own end-to-end acceptance tests) on the CPU emulation: the
1-D ODE with the reference's three-stage Adam schedule under all five training strategies (resident-theta loop, device samplers,
minibatch designs, quadrature stand-in) and the Flux-translation test — the same statements and tolerances, seconds each."""
import pytest

import test_gpu_reference_acceptance as acc


@pytest.mark.parametrize("strategy", ["grid", "stochastic", "quasirandom_minibatch", "quasirandom_resampling", "quadrature"])
def test_simple_1d_ode_all_strategies_on_emulation(npde, use_emu, emu_lib, strategy):
    acc.test_simple_1d_ode_all_strategies(npde, emu_lib, strategy)


def test_translating_from_flux_on_emulation(npde, use_emu, emu_lib):
    acc.test_translating_from_flux(npde, emu_lib)
