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
// stage 1: entry e (a slab offset, or a loss column for e >= nent) summed over one contiguous chunk of workgroups
AUX_DEV void reduce1_body(int e, int chunk, double* tmp, const float* slabs, int slab, int nblocks, int nsplit, const int* ent_off,
                          int nent, const double* losspart, int K) {
    const int per = (nblocks + nsplit - 1) / nsplit;
    const int b0 = chunk * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
    double s = 0.0;
    if (e < nent) {
        const int so = ent_off[e];
        for (int b = b0; b < b1; ++b) s += (double)slabs[(size_t)b * slab + so];
    } else {
        const int k = e - nent;
        for (int wv = b0 * 4; wv < b1 * 4; ++wv) s += losspart[(size_t)wv * K + k];
    }
    tmp[(size_t)chunk * (nent + K) + e] = s;
}
// stage 2: theta row r = sum over its entries and over the chunks, in a fixed order
AUX_DEV void reduce2_body(int r, double* gradd, double* lossraw, const double* tmp, int nsplit, const int* row_ptr, const int* row_theta,
                          int nrows, int nent, int K) {
    if (r < nrows) {
        double s = 0.0;
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e)
            for (int ch = 0; ch < nsplit; ++ch) s += tmp[(size_t)ch * (nent + K) + e];
        gradd[row_theta[r]] += s;
    } else {
        const int k = r - nrows;
        double s = 0.0;
        for (int ch = 0; ch < nsplit; ++ch) s += tmp[(size_t)ch * (nent + K) + nent + k];
        lossraw[k] += s;
    }
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
struct ReduceArgs {
    double* gradd; double* lossraw; double* tmp; const float* slabs; int slab; int nblocks; int nsplit;
    const int* ent_off; int nent; const int* row_ptr; const int* row_theta; int nrows; const double* losspart; int K;
};
inline void launch_reduce(const ReduceArgs& a, plat_stream) {
    for (int ch = 0; ch < a.nsplit; ++ch)
        for (int e = 0; e < a.nent + a.K; ++e)
            reduce1_body(e, ch, a.tmp, a.slabs, a.slab, a.nblocks, a.nsplit, a.ent_off, a.nent, a.losspart, a.K);
    for (int r = 0; r < a.nrows + a.K; ++r)
        reduce2_body(r, a.gradd, a.lossraw, a.tmp, a.nsplit, a.row_ptr, a.row_theta, a.nrows, a.nent, a.K);
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
struct ReduceArgs {
    double* gradd; double* lossraw; double* tmp; const float* slabs; int slab; int nblocks; int nsplit;
    const int* ent_off; int nent; const int* row_ptr; const int* row_theta; int nrows; const double* losspart; int K;
};
__global__ void k_reduce1(const ReduceArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < a.nent + a.K) reduce1_body(e, (int)blockIdx.y, a.tmp, a.slabs, a.slab, a.nblocks, a.nsplit, a.ent_off, a.nent, a.losspart, a.K);
}
__global__ void k_reduce2(const ReduceArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < a.nrows + a.K) reduce2_body(r, a.gradd, a.lossraw, a.tmp, a.nsplit, a.row_ptr, a.row_theta, a.nrows, a.nent, a.K);
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
inline void launch_reduce(const ReduceArgs& a, plat_stream st) {
    const int n1 = a.nent + a.K, n2 = a.nrows + a.K;
    hipLaunchKernelGGL(k_reduce1, dim3((n1 + 255) / 256, a.nsplit), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_reduce2, dim3((n2 + 255) / 256), dim3(256), 0, st, a);
}
inline void launch_finish(float* out, const double* gradd, const double* lossraw, int P, int K, plat_stream st) {
    const int n = P + K;
    hipLaunchKernelGGL(k_finish, dim3((n + 255) / 256), dim3(256), 0, st, out, gradd, lossraw, P, K);
}
#endif

}  // namespace aux
