// micro-benchmark: MFMA issue time of an fp32 16x16x64 product as (a) 16 x v_mfma_f32_16x16x4_f32, (b) 12 x v_mfma_f32_16x16x32_bf16
// (3-way bf16 split of both operands, 6 cross terms, K = 64 = 2 x 32), operands in registers, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, const float* in, int iters) {
    const int l = threadIdx.x;
    f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float a = in[l], b = in[l + 64];
    bf8 ah, am, al, bh, bm, bl;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); am[i] = (__bf16)(a * 0.01f + i); al[i] = (__bf16)(a * 1e-4f); bh[i] = (__bf16)(b + i); bm[i] = (__bf16)(b * 0.01f); bl[i] = (__bf16)(b * 1e-4f); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[q], 0, 0, 0);
                }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (float)(t1 - t0);
}
int main() {
    float *out, *in; int nb = 512;
    hipMalloc(&out, (nb * 256 + 1) * 4); hipMalloc(&in, 1024);
    hipMemset(in, 0, 1024);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, in, 2000); else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, in, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float cyc; hipMemcpy(&cyc, out + nb * 256, 4, hipMemcpyDeviceToHost);
            printf("mode %d (%s): %.3f ms, %.0f memtime ticks per 4 output tiles x K=64 -> %.1f per tile  (2 waves/SIMD, 512 WGs)\n", mode, mode ? "12 x bf16 16x16x32" : "16 x f32 16x16x4", ms, cyc / 2000, cyc / 2000 / 4);
        }
    }
    return 0;
}
