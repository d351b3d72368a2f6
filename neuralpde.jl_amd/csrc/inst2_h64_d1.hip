// widths 33..64, 1 input (ODE-like problems), 2 to 4 hidden layers: value-only and {u, u', u''}
#include "spec_registry.hpp"
PINN_INSTANTIATE2(f2_h64n1d1_val, 64, 1, 1, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n1d1_lap, 64, 1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE2(f2_h64n2d1_val, 64, 2, 1, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n2d1_lap, 64, 2, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE2(f2_h64n3d1_val, 64, 3, 1, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n3d1_lap, 64, 3, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
