// update_rules.hpp — the optimiser arithmetic shared by the stand-alone update kernels (aux_kernels.hpp: k_adam, k_adam_fused, k_adam_dev)
// and the persistent training kernel (pinn_train.hpp): ONE statement of the rule, so that K iterations inside one launch and K single steps
// round identically.  Free of host headers (the kernel translation units and the hiprtc back end include it).
#pragma once

#if defined(PINN_EMU)
#define UR_DEV inline
#include <cmath>
#else
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#define UR_DEV __device__ __forceinline__
#endif

namespace ur {

// Adam exactly as [3P] Optimisers.Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; theta -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps).
// The multiply-adds are spelled out as fused operations, so that every kernel that performs the update and the CPU emulation round
// identically — left to the compiler, two kernels with the same source expression contracted it differently (1-ulp differences from the
// second step on).  c1 = 1/(1-b1^t), c2 = 1/(1-b2^t).
UR_DEV float adam_update(float th, float& m, float& v, float g, float lr, float b1, float b2, float eps, float c1, float c2) {
    m = __builtin_fmaf(b1, m, (1.0f - b1) * g);
    v = __builtin_fmaf(b2, v, ((1.0f - b2) * g) * g);
    return th - (lr * (m * c1)) / (sqrtf(v * c2) + eps);
}

}  // namespace ur
