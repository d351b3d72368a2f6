// 1-D, 3x32-class nets (BASELINE config 1): {u, u_x, u_xx} and value-only
#include "spec_registry.hpp"
PINN_INSTANTIATE(h32n2d1_lap, 32, 2, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE(h32n2d1_val, 32, 2, 1, 0x0, 0ull, 0, 5)
