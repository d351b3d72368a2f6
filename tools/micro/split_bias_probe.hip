// r05: rounding behaviour of the split-operand product (six v_mfma_f32_16x16x32_bf16 per K = 32 block) against an fp32 MFMA chain and double:
// mean SIGNED error (a bias acts coherently on every pre-activation of a network — what a trained theta amplifies) and rms error of
// z = sum_k W[i][k] a[k][j] over K = 64, for the engine's order of the six pieces, the reverse order, and the small pieces in their own accumulator.
//   hipcc -O3 -w --offload-arch=gfx950 tools/micro/split_bias_probe.hip -o tools/micro/split_bias_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; const float r = x - (float)h; m = (__bf16)r; const float r2 = r - (float)m; l = (__bf16)r2;
}
// W: [tiles][16][64] row-major, A: [tiles][64][16]; out: [tiles][mode][16][16]
__global__ void __launch_bounds__(64) k(const float* W, const float* A, const float* C0, float* out, int nmodes) {
    const int l = threadIdx.x, tile = blockIdx.x, g = l >> 4, c = l & 15;
    const float* w = W + (size_t)tile * 16 * 64;
    const float* a = A + (size_t)tile * 64 * 16;
    f4 r[5];
    for (int m = 0; m < 5; ++m) for (int e = 0; e < 4; ++e) r[m][e] = C0[((size_t)tile * 16 + 4 * g + e) * 16 + c];
    f4 small = {0, 0, 0, 0};
    // fp32 MFMA chain: 16 x (16x16x4)
    for (int k4 = 0; k4 < 16; ++k4) r[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c * 64 + 4 * k4 + g], a[(4 * k4 + g) * 16 + c], r[0], 0, 0, 0);
    for (int kb = 0; kb < 2; ++kb) {
        bf8 ah, am, al, bh, bm, bl;
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, lo;
            split3(w[c * 64 + 32 * kb + 8 * g + e], h, m, lo); ah[e] = h; am[e] = m; al[e] = lo;
            split3(a[(32 * kb + 8 * g + e) * 16 + c], h, m, lo); bh[e] = h; bm[e] = m; bl[e] = lo;
        }
        // 1: the engine's order
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, r[1], 0, 0, 0);
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, r[1], 0, 0, 0);
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, r[1], 0, 0, 0);
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, r[1], 0, 0, 0);
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, r[1], 0, 0, 0);
        r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, r[1], 0, 0, 0);
        // 2: reverse order (smallest first)
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, r[2], 0, 0, 0);
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, r[2], 0, 0, 0);
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, r[2], 0, 0, 0);
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, r[2], 0, 0, 0);
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, r[2], 0, 0, 0);
        r[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, r[2], 0, 0, 0);
        // 3: the five small pieces in their own accumulator (smallest first), added once at the end
        small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, small, 0, 0, 0);
        small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, small, 0, 0, 0);
        small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, small, 0, 0, 0);
        small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, small, 0, 0, 0);
        small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, small, 0, 0, 0);
        r[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, r[3], 0, 0, 0);
        // 4: hh only (what the dropped pieces are worth)
        r[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, r[4], 0, 0, 0);
    }
    for (int e = 0; e < 4; ++e) r[3][e] += small[e];
    for (int m = 0; m < nmodes; ++m) for (int e = 0; e < 4; ++e) out[(((size_t)tile * nmodes + m) * 16 + 4 * g + e) * 16 + c] = r[m][e];
}
int main() {
    const int T = 4096, NM = 5;
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> nw(0.f, 0.35f);
    std::uniform_real_distribution<float> ua(-1.f, 1.f);
    for (int pass = 0; pass < 4; ++pass) {
        std::vector<float> W((size_t)T * 16 * 64), A((size_t)T * 64 * 16), C0((size_t)T * 256, 0.f);
        for (auto& x : W) x = nw(rng);
        for (auto& x : A) x = pass == 1 ? std::tanh(2.5f * ua(rng)) : ua(rng);                     // pass 1: saturating activations
        if (pass == 2) for (auto& x : C0) x = 3.0f * ua(rng);                                      // pass 2: a large initial accumulator (bias / previous block)
        if (pass == 3) for (auto& x : C0) x = 200.0f * ua(rng);                                    // pass 3: a running sum 100 x the block's products (dW accumulators after ~100 tiles)
        float *dW, *dA, *dC, *dO;
        hipMalloc(&dW, W.size() * 4); hipMalloc(&dA, A.size() * 4); hipMalloc(&dC, C0.size() * 4); hipMalloc(&dO, (size_t)T * NM * 256 * 4);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C0.data(), C0.size() * 4, hipMemcpyHostToDevice);
        k<<<T, 64>>>(dW, dA, dC, dO, NM);
        std::vector<float> O((size_t)T * NM * 256);
        hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
        const char* names[NM] = {"fp32 MFMA chain (16 x 16x16x4)", "split x6, engine order (hh hm mh hl mm lh)", "split x6, reverse order", "split: hh + own accumulator for the 5 small", "hh only"};
        std::printf("pass %d (%s): errors of z = C0 + sum_k W a, K = 64, in units of 1e-8 (|z| rms ~ 1.6)\n", pass, pass == 0 ? "a uniform(-1,1)" : (pass == 1 ? "a = tanh(2.5 u)" : (pass == 2 ? "a uniform, C0 = 3 u" : "a uniform, C0 = 200 u: ulp(C0) ~ 1.5e-5")));
        for (int m = 0; m < NM; ++m) {
            double sum = 0, sumsq = 0, sumrel = 0, sumsgn = 0; size_t n = 0;
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < 16; ++i)
                    for (int j = 0; j < 16; ++j) {
                        double ex = C0[((size_t)t * 16 + i) * 16 + j];
                        for (int kk = 0; kk < 64; ++kk) ex += (double)W[((size_t)t * 16 + i) * 64 + kk] * (double)A[((size_t)t * 64 + kk) * 16 + j];
                        const double e = (double)O[(((size_t)t * NM + m) * 16 + i) * 16 + j] - ex;
                        sum += e; sumsq += e * e; sumrel += e / (std::fabs(ex) + 1e-30) * (std::fabs(ex) > 0.1); sumsgn += e * (ex > 0 ? 1 : -1); ++n;
                    }
            std::printf("   %-46s mean %+8.3f   mean toward-larger-|z| %+8.3f   rms %8.3f\n", names[m], sum / n * 1e8, sumsgn / n * 1e8, std::sqrt(sumsq / n) * 1e8);
        }
        hipFree(dW); hipFree(dA); hipFree(dC); hipFree(dO);
    }
    return 0;
}
