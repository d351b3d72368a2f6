// MERGED launches (wave_main2m), 2-D 4x64-class nets: the interior term's jet set + the value-only set of the boundary terms in one
// persistent kernel (BASELINE config 2: forward-Laplacian set; config 3: Burgers set; the plain {u, u_x, u_y, u_xx, u_yy} set)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_PAIR(f2m_h64n3d2_lapc_val, 64, 3, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3), 0x0, 0ull, 0, 4, 0u)
PINN_INSTANTIATE2_PAIR(f2m_h64n3d2_burg_val, 64, 3, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1, 0u, 0x0, 0ull, 0, 4, 0u)
PINN_INSTANTIATE2_PAIR(f2m_h64n3d2_lap_val, 64, 3, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1, 0u, 0x0, 0ull, 0, 4, 0u)
