// micro-probe: semantics and bank behaviour of gfx950's LDS transpose read ds_read_b64_tr_b16 (the dW operands of the split-operand
// kernels are read with it straight out of the B-operand exchange images, csrc/pinn_kernels2.hpp: dw_pair_tr).
//   part A: which LDS halfword lands in which (lane, element) — checked against the hypothesis
//           out[16 G + n][j] = lds[addr(16 G + 4 j + n / 4) + n % 4]          (addr in halfwords, a 16-lane group transposes a [4][16] block)
//   part B: cycles per wave-instruction for the address patterns the kernel could use (one wave per SIMD, no other LDS traffic).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/tr_probe.hip -o tools/micro/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4* lds_s4;

__global__ void k_sem(short* out, const int* addr) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

// pattern p: byte address of lane l
__device__ int pat(int p, int l, int jh, int h) {
    const int g2 = l >> 4, i = l & 15, kq = i >> 2, gg = i & 3;
    const int pc = 8 * (g2 & 1) + 4 * jh + kq;
    switch (p) {
        case 0: return l * 8;                                               // linear
        case 1: return (16 * gg + pc) * 16 + 8 * h + (g2 >> 1) * 1024;      // today's image: [lane][8 bf16], 16-byte slots
        case 2: return (16 * gg + pc) * 8 + (g2 >> 1) * 512;                // plane image [h][lane][4 bf16], unswizzled
        case 3: return (16 * gg + (pc ^ (4 * (gg >> 1)))) * 8 + (g2 >> 1) * 512;   // plane image, points swizzled by 4 (gg >> 1)
        case 4: return (16 * gg + (pc ^ (8 * (gg >> 1)))) * 8 + (g2 >> 1) * 512;   // swizzled by 8 (gg >> 1)
        case 5: return (16 * gg + (pc ^ (2 * (gg >> 1)))) * 8 + (g2 >> 1) * 512;   // swizzled by 2 (gg >> 1)
        default: return (16 * gg + (pc ^ gg)) * 8 + (g2 >> 1) * 512;              // swizzled by gg
    }
}

template <int KIND>   // 0: tr_b16, 1: plain ds_read_b64, 2: ds_read_b128
__global__ void __launch_bounds__(256, 2) k_time(float* out, int p, int iters) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x & 63;
    unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) short*)lds;
    unsigned ad[4];
    for (int jh = 0; jh < 2; ++jh)
        for (int h = 0; h < 2; ++h) ad[jh * 2 + h] = a0 + (KIND == 2 ? l * 16 : pat(p, l, jh, h));
    int acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int r[64];
        if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(int4*)&r[4 * u]) : "v"(ad[u & 3]), "n"(0));
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(*(int2*)&r[2 * u]) : "v"(ad[u & 3]), "n"(0));
                else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(int2*)&r[2 * u]) : "v"(ad[u & 3]), "n"(0));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 64; ++u) acc ^= r[u];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (float)(t1 - t0);
}

int main() {
    // ---- part A
    short* out; int* addr;
    hipMalloc(&out, 64 * 4 * 2); hipMalloc(&addr, 64 * 4);
    int h_addr[64]; short h_out[256];
    for (int test = 0; test < 2; ++test) {
        srand(7);
        for (int l = 0; l < 64; ++l) h_addr[l] = test == 0 ? 4 * l : 4 * (rand() % 2048);
        hipMemcpy(addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, out, addr);
        hipMemcpy(h_out, out, sizeof h_out, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int G = l >> 4, n = l & 15;
                const int expect = h_addr[16 * G + 4 * j + n / 4] + n % 4;
                if (h_out[l * 4 + j] != (short)expect) ++bad;
            }
        printf("part A test %d (%s addresses): %d of 256 elements differ from out[16G+n][j] = lds[addr(16G + 4j + n/4) + n%%4]\n", test, test ? "random" : "linear", bad);
        if (test == 0 || bad) {
            for (int l = 0; l < 20; ++l) printf("  lane %2d (addr %4d): %5d %5d %5d %5d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
        }
    }
    // ---- part B
    float* tout; const int nb = 512;
    hipMalloc(&tout, (nb * 256 + 1) * 4);
    const char* names[] = {"linear lane*8", "today's [lane][8] image, fixed half", "plane image, unswizzled", "plane, pc ^ 4(gg>>1)", "plane, pc ^ 8(gg>>1)", "plane, pc ^ 2(gg>>1)", "plane, pc ^ gg"};
    for (int kind = 0; kind < 3; ++kind)
        for (int p = 0; p < (kind == 2 ? 1 : 7); ++p) {
            float cyc = 0;
            for (int rep = 0; rep < 2; ++rep) {
                const int iters = 4000;
                if (kind == 0) hipLaunchKernelGGL(k_time<0>, dim3(nb), dim3(256), 0, 0, tout, p, iters);
                else if (kind == 1) hipLaunchKernelGGL(k_time<1>, dim3(nb), dim3(256), 0, 0, tout, p, iters);
                else hipLaunchKernelGGL(k_time<2>, dim3(nb), dim3(256), 0, 0, tout, p, iters);
                hipDeviceSynchronize();
                hipMemcpy(&cyc, tout + nb * 256, 4, hipMemcpyDeviceToHost);
                cyc /= iters;
            }
            printf("part B %-18s %-40s %7.1f ticks per iteration of %d reads per wave (8 waves per CU, all reads of an iteration in flight)\n", kind == 0 ? "ds_read_b64_tr_b16" : (kind == 1 ? "ds_read_b64" : "ds_read_b128"), kind == 2 ? "linear lane*16" : names[p], cyc, kind == 2 ? 16 : 32);
        }
    return 0;
}
