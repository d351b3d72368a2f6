// r05: float64 tanh for the matrix-pipe float64 kernels (csrc/pinn_kernels5.hpp): ocml tanh(double) against exp-based forms, accuracy vs long double
// and cycles per evaluation.   hipcc -O3 -w --offload-arch=gfx950 tools/micro/tanh64_probe.hip -o tools/micro/tanh64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double exp_nonpos(double y) {           // e^y, -80 <= y <= 0: y = n ln2 + r, degree-13 Taylor of e^r on |r| <= ln2 / 2
    const double n = __builtin_rint(y * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, y);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                              // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
template <int V> __device__ __forceinline__ double tanh_v(double x) {
    if (V == 0) return tanh(x);
    const double ax = __builtin_fmin(__builtin_fabs(x), 40.0);
    const double e = (V == 1) ? exp(-2.0 * ax) : exp_nonpos(-2.0 * ax);
    const double s = 1.0 + e, d = 1.0 - e;
    double t;
    if (V == 3) {
        double r = __builtin_amdgcn_rcp(s);
        r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
        t = d * r;
        t = __builtin_fma(__builtin_fma(-s, t, d), r, t);
    } else t = d / s;
    return __builtin_copysign(t, x);
}
template <int V> __global__ void k_eval(const double* x, double* y, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = tanh_v<V>(x[i]); }
template <int V> __global__ void k_time(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = a + 0.3, c = a + 0.7, d = a - 0.4;
    for (int i = 0; i < iters; ++i) { a = tanh_v<V>(a + 0.1); b = tanh_v<V>(b + 0.1); c = tanh_v<V>(c - 0.1); d = tanh_v<V>(d - 0.2); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int V> void run(const std::vector<double>& hx, double* dx, double* dy, const char* name) {
    const int n = (int)hx.size();
    k_eval<V><<<(n + 255) / 256, 256>>>(dx, dy, n);
    std::vector<double> hy(n);
    hipMemcpy(hy.data(), dy, n * sizeof(double), hipMemcpyDeviceToHost);
    double maxabs = 0, maxrel = 0, sumsq = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = tanhl((long double)hx[i]);
        const double e = (double)((long double)hy[i] - t);
        maxabs = std::fmax(maxabs, std::fabs(e)); sumsq += e * e;
        if (t != 0) maxrel = std::fmax(maxrel, std::fabs(e / (double)t));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_time<V><<<1024, 64>>>(dy, 16);
    hipEventRecord(e0); k_time<V><<<1024, 64>>>(dy, 2048); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * 2048);      // one wave per SIMD, 4 x 2048 evaluations each
    std::printf("%-44s max|err| %.3e  rms %.3e  max rel %.3e   ~%.0f cycles per evaluation (one wave per SIMD)\n", name, maxabs, std::sqrt(sumsq / n), maxrel, cyc);
}
int main() {
    std::vector<double> hx;
    for (int i = 0; i < (1 << 21); ++i) { const double u = (i + 0.5) / (1 << 21); hx.push_back(-10.0 + 20.0 * u); }
    for (int i = 0; i < (1 << 19); ++i) { const double u = (i + 0.5) / (1 << 19); hx.push_back(-0.05 + 0.1 * u); }
    double *dx, *dy; hipMalloc(&dx, hx.size() * 8); hipMalloc(&dy, hx.size() * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
    std::printf("float64 tanh against long double, %zu arguments in [-10, 10] (+ a dense set around 0); ulp(1) = 2.2e-16\n", hx.size());
    run<0>(hx, dx, dy, "0: ocml tanh");
    run<1>(hx, dx, dy, "1: (1-e)/(1+e), ocml exp, division");
    run<2>(hx, dx, dy, "2: (1-e)/(1+e), own exp, division");
    run<3>(hx, dx, dy, "3: (1-e)/(1+e), own exp, rcp + 2 Newton + fix");
    return 0;
}
