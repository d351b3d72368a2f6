// micro-benchmark: issue cost (shader cycles per wave64 instruction) of the VALU instructions that dominate the element-wise phases of the
// residual kernels — v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_cvt_pk_bf16_f32, v_sub_f32 — with one and with two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[8], b = out[threadIdx.x] + 1.0f, c = 0.5f;
    for (int i = 0; i < 8; ++i) a[i] = b + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8]; for (int i = 0; i < 8; ++i) p[i] = f2{b + i, b - i};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (float)(t1 - t0);
}
int main() {
    float* out; hipMalloc(&out, (512 * 256 + 1) * 4); hipMemset(out, 0, (512 * 256 + 1) * 4);
    const char* nm[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_sub_f32", "v_and_b32", "v_lshlrev_b32"};
    for (int nb = 256; nb <= 512; nb += 256)
        for (int kind = 0; kind < 8; ++kind) {
            float cyc = 0; const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                    default: hipLaunchKernelGGL(k<7>, dim3(nb), dim3(256), 0, 0, out, iters); break;
                }
                hipDeviceSynchronize();
                hipMemcpy(&cyc, out + nb * 256, 4, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s) per SIMD  %-20s %6.2f ticks per wave-instruction (this wave's clock; 32 independent-ish instructions per iteration)\n", nb / 256, nm[kind], cyc / iters / 32);
        }
    return 0;
}
