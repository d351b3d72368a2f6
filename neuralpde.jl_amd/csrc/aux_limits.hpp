// aux_limits.hpp — size limits shared by the auxiliary kernels (aux_kernels.hpp) and the host-side planner.
#pragma once
namespace aux {
constexpr int MAX_PACK_NETS = 16;     // networks whose packed images the fused optimiser update can address
constexpr int MAX_GROUPS = 24;        // launch groups + coupled pseudo-groups per handle (kernel-argument arrays of the reduction)
constexpr int EXPR_MAX_SLOTS = 24;    // jet slots of one coupled equation (k_expr)
constexpr int EXPR_MAX_ROWS = 96;     // tape rows of k_expr / k_src (per-thread arrays)
constexpr int SRC_MAX = 8;            // coordinate-only source channels per term
}  // namespace aux
