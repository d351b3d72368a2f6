// family 2 (neuron-split workgroups, 2 workgroups per CU): 2-D, 4x64-class nets
#include "spec_registry.hpp"
PINN_INSTANTIATE2(f2_h64n3d2_lap, 64, 3, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
#ifndef PINN_VAL_PG
#define PINN_VAL_PG 4
#endif
PINN_INSTANTIATE2_HI_SIN(f2_h64n3d2_val, 64, 3, 2, 0x0, 0ull, 0, PINN_VAL_PG, 0u)
PINN_INSTANTIATE2(f2_h64n3d2_burg, 64, 3, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1)
