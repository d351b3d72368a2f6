// MERGED launches (wave_main2m), 4-D 128-wide nets: forward-Laplacian heat set + value-only set (BASELINE config 5: 6 x 128; 2 x 128 for
// the unit tests)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_PAIR(f2m_h128n5d4_lapc_val, 128, 5, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE), 0x0, 0ull, 0, 4, 0u)
PINN_INSTANTIATE2_PAIR(f2m_h128n1d4_lapc_val, 128, 1, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE), 0x0, 0ull, 0, 4, 0u)
