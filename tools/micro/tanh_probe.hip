// r05: accuracy (against double tanh) and issue cost of the tanh formulas of csrc/vec.hpp (PINN_ACT_TANH = 0 .. 3) on the device, plus ocml tanhf.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/tanh_probe.hip -o tools/micro/tanh_probe && tools/micro/tanh_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
template <int V> __device__ __forceinline__ float tanh_v(float x) {
    if (V == 4) return tanhf(x);
    if (V == 0) {
        const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
        return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    }
    const float ax = __builtin_fabsf(x);
    const float e = (V >= 3) ? __builtin_amdgcn_exp2f(__builtin_fmaf(ax, -2.885390043258667f, ax * -3.851926067000022e-08f))
                             : __builtin_amdgcn_exp2f(ax * -2.8853900817779268f);
    const float s = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(s);
    if (V >= 2) r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
    return __builtin_copysignf((1.0f - e) * r, x);
}
template <int V> __global__ void k_eval(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = tanh_v<V>(x[i]);
}
template <int V> __global__ void k_time(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = a + 0.3f, c = a + 0.7f, d = a - 0.4f;
    for (int i = 0; i < iters; ++i) { a = tanh_v<V>(a + 0.1f); b = tanh_v<V>(b + 0.1f); c = tanh_v<V>(c - 0.1f); d = tanh_v<V>(d - 0.2f); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int V> void run(const std::vector<float>& hx, float* dx, float* dy, const char* name) {
    const int n = (int)hx.size();
    k_eval<V><<<(n + 255) / 256, 256>>>(dx, dy, n);
    std::vector<float> hy(n);
    hipMemcpy(hy.data(), dy, n * sizeof(float), hipMemcpyDeviceToHost);
    double maxabs = 0, sumsq = 0, maxrel = 0, sumsigned = 0, maxabs_small = 0; double argmax = 0;
    for (int i = 0; i < n; ++i) {
        const double t = std::tanh((double)hx[i]), e = (double)hy[i] - t;
        if (std::fabs(e) > maxabs) { maxabs = std::fabs(e); argmax = hx[i]; }
        sumsq += e * e; sumsigned += e * (hx[i] >= 0 ? 1 : -1);
        if (t != 0) maxrel = std::fmax(maxrel, std::fabs(e / t));
        if (std::fabs(hx[i]) < 0.25) maxabs_small = std::fmax(maxabs_small, std::fabs(e));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_time<V><<<1024, 256>>>(dy, 16);
    hipEventRecord(e0); k_time<V><<<1024, 256>>>(dy, 4096); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1024 blocks x 4 waves / (256 CUs x 4 SIMDs) = 4 waves per SIMD, 4 x 4096 dependent-free-ish evaluations each
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * 4 * 4096);
    std::printf("%-28s max|err| %.3e (at x = %+.4f)  rms %.3e  mean signed (odd part) %+.2e  max|err| for |x|<0.25 %.3e  max rel %.3e  ~%.1f cycles per evaluation and SIMD\n",
                name, maxabs, argmax, std::sqrt(sumsq / n), sumsigned / n, maxabs_small, maxrel, cyc);
}
int main() {
    std::vector<float> hx;
    for (int i = 0; i < (1 << 22); ++i) { const double u = (i + 0.5) / (1 << 22); hx.push_back((float)(-6.0 + 12.0 * u)); }
    for (int i = 0; i < (1 << 20); ++i) { const double u = (i + 0.5) / (1 << 20); hx.push_back((float)(-0.25 + 0.5 * u)); }
    float *dx, *dy; hipMalloc(&dx, hx.size() * 4); hipMalloc(&dy, hx.size() * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::printf("tanh formulas against double tanh, %zu arguments in [-6, 6] (+ a dense set in [-0.25, 0.25]); ulp(1) = 1.19e-7\n", hx.size());
    run<0>(hx, dx, dy, "0: 1 - 2/(e^{2x}+1) (r01-r04)");
    run<1>(hx, dx, dy, "1: odd (1-e)/(1+e), rcp");
    run<2>(hx, dx, dy, "2: odd + Newton on rcp");
    run<3>(hx, dx, dy, "3: 2 + two-constant exponent");
    run<4>(hx, dx, dy, "4: ocml tanhf");
    return 0;
}
