// aux_kernels.hpp — small HBM-side kernels around the fused residual kernel.
//   k_pack   : theta (ComponentArrays order, src/discretize.jl:451-465) -> padded MFMA-fragment order
//   k_params : theta.p / default_p -> parameter rows of the residual tape (src/discretize.jl:83-109)
//   k_reduce : fixed-order sum of the per-wave gradient slabs and loss partials (deterministic)
//   k_finish : double accumulators -> [P floats grad | K floats raw sums]
#pragma once
#include "plat.hpp"

namespace aux {

#ifdef PINN_EMU
#define AUX_DEV inline
#else
#define AUX_DEV __device__ __forceinline__
#endif

AUX_DEV void pack_body(int i, float* packed, const int* idx, const float* theta) {
    const int j = idx[i];
    packed[i] = (j >= 0) ? theta[j] : 0.f;
}
AUX_DEV void params_body(int j, float* params, const float* theta, const float* defaults, int ne, int p_off) {
    params[j] = (j < ne) ? theta[p_off + j] : defaults[j];
}
AUX_DEV void reduce_body(int e, double* gradd, const float* slabs, int slab, int nwaves, const int* map_theta, const int* map_slab) {
    const int so = map_slab[e];
    double s = 0.0;
    for (int w = 0; w < nwaves; ++w) s += (double)slabs[(size_t)w * slab + so];
    gradd[map_theta[e]] += s;
}
AUX_DEV void reduce_loss_body(int k, double* lossraw, const double* losspart, int nwaves, int K) {
    double s = 0.0;
    for (int w = 0; w < nwaves; ++w) s += losspart[(size_t)w * K + k];
    lossraw[k] += s;
}
AUX_DEV void finish_body(int i, float* out, const double* gradd, const double* lossraw, int P, int K) {
    if (i < P) out[i] = (float)gradd[i];
    else if (i < P + K) out[i] = (float)lossraw[i - P];
}

#ifdef PINN_EMU
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream) {
    for (int i = 0; i < n; ++i) pack_body(i, packed, idx, theta);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream) {
    for (int j = 0; j < np; ++j) params_body(j, params, theta, defaults, ne, p_off);
}
inline void launch_reduce(double* gradd, double* lossraw, const float* slabs, int slab, int nwaves, const int* mt, const int* ms, int nmap,
                          const double* losspart, int K, plat_stream) {
    for (int e = 0; e < nmap; ++e) reduce_body(e, gradd, slabs, slab, nwaves, mt, ms);
    for (int k = 0; k < K; ++k) reduce_loss_body(k, lossraw, losspart, nwaves, K);
}
inline void launch_finish(float* out, const double* gradd, const double* lossraw, int P, int K, plat_stream) {
    for (int i = 0; i < P + K; ++i) finish_body(i, out, gradd, lossraw, P, K);
}
#else
__global__ void k_pack(float* packed, const int* idx, const float* theta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pack_body(i, packed, idx, theta);
}
__global__ void k_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off) {
    const int j = threadIdx.x;
    if (j < np) params_body(j, params, theta, defaults, ne, p_off);
}
__global__ void k_reduce(double* gradd, double* lossraw, const float* slabs, int slab, int nwaves, const int* mt, const int* ms, int nmap,
                         const double* losspart, int K) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nmap) reduce_body(e, gradd, slabs, slab, nwaves, mt, ms);
    else if (e < nmap + K) reduce_loss_body(e - nmap, lossraw, losspart, nwaves, K);
}
__global__ void k_finish(float* out, const double* gradd, const double* lossraw, int P, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    finish_body(i, out, gradd, lossraw, P, K);
}
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream st) {
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, st, packed, idx, theta, n);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream st) {
    if (np > 0) hipLaunchKernelGGL(k_params, dim3(1), dim3(64), 0, st, params, theta, defaults, np, ne, p_off);
}
inline void launch_reduce(double* gradd, double* lossraw, const float* slabs, int slab, int nwaves, const int* mt, const int* ms, int nmap,
                          const double* losspart, int K, plat_stream st) {
    const int n = nmap + K;
    hipLaunchKernelGGL(k_reduce, dim3((n + 63) / 64), dim3(64), 0, st, gradd, lossraw, slabs, slab, nwaves, mt, ms, nmap, losspart, K);
}
inline void launch_finish(float* out, const double* gradd, const double* lossraw, int P, int K, plat_stream st) {
    const int n = P + K;
    hipLaunchKernelGGL(k_finish, dim3((n + 255) / 256), dim3(256), 0, st, out, gradd, lossraw, P, K);
}
#endif

}  // namespace aux
