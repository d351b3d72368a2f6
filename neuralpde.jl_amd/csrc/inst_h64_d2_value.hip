// 2-D, 4x64-class nets: value-only jet set {u} (Dirichlet BC terms, phi inference), 80 points per wave tile
#include "spec_registry.hpp"
PINN_INSTANTIATE(h64n3d2_val, 64, 3, 2, 0x0, 0ull, 0, 5)
