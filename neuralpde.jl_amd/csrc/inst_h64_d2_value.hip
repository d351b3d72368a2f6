// 2-D, 4x64-class nets: value-only jet set {u} (Dirichlet BC terms, phi inference), 64 points per wave tile (4 column groups: 65,536-point terms tile into exactly 1024 wave tiles)
#include "spec_registry.hpp"
PINN_INSTANTIATE(h64n3d2_val, 64, 3, 2, 0x0, 0ull, 0, 4)
