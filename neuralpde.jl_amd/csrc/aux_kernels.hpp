// aux_kernels.hpp — small HBM-side kernels around the fused residual kernel.
//   k_pack    : theta (ComponentArrays order, src/discretize.jl:451-465) -> padded MFMA-fragment order
//   k_params  : theta.p / default_p -> parameter rows of the residual tape (src/discretize.jl:83-109)
//   k_reduce1 : stage 1 of the fixed-order reduction: every gradient-slab offset (dense, coalesced) and loss column of
//               every launch group summed over one contiguous chunk of workgroups
//   k_reduce2 : stage 2: theta element p = sum over the groups / slab entries / chunks that feed it, written straight
//               into the output vector [P floats grad | K floats raw sums of squares] (+ K doubles for the host path)
// Every sum has a fixed order => bit-identical results run to run.
#pragma once
#include "plat.hpp"

namespace aux {

#ifdef PINN_EMU
#define AUX_DEV inline
#else
#define AUX_DEV __device__ __forceinline__
#endif

constexpr int MAX_GROUPS = 8;

struct Reduce1Args {
    double* tmp[MAX_GROUPS];            // [nsplit][nent + K]
    const float* slabs[MAX_GROUPS];     // [nblocks][slab]
    const double* losspart[MAX_GROUPS]; // [nblocks*4][K]
    int slab[MAX_GROUPS], nblocks[MAX_GROUPS], nsplit[MAX_GROUPS], nent[MAX_GROUPS], active[MAX_GROUPS];
    int K;
};
struct Reduce2Args {
    float* out;                         // [P + K]
    double* lossraw;                    // [K] (nullable)
    const int* row_ptr;                 // [P + 1]
    const int* row_grp;                 // group of every contribution
    const int* row_ent;                 // slab-entry index (within its group) of every contribution
    const double* tmp[MAX_GROUPS];
    int stride[MAX_GROUPS], nsplit[MAX_GROUPS], nent[MAX_GROUPS], active[MAX_GROUPS];
    int ngroups, P, K;
};

AUX_DEV void pack_body(int i, float* packed, const int* idx, const float* theta) {
    const int j = idx[i];
    packed[i] = (j >= 0) ? theta[j] : 0.f;
}
AUX_DEV void params_body(int j, float* params, const float* theta, const float* defaults, int ne, int p_off) {
    params[j] = (j < ne) ? theta[p_off + j] : defaults[j];
}
AUX_DEV void reduce1_body(int e, int chunk, int g, const Reduce1Args& a) {
    if (!a.active[g] || chunk >= a.nsplit[g] || e >= a.nent[g] + a.K) return;
    const int nb = a.nblocks[g], ns = a.nsplit[g];
    const int per = (nb + ns - 1) / ns;
    const int b0 = chunk * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    double s = 0.0;
    if (e < a.nent[g]) {      // dense over slab offsets: consecutive threads read consecutive floats of every slab
        const float* p = a.slabs[g] + e;
        for (int b = b0; b < b1; ++b) s += (double)p[(size_t)b * a.slab[g]];
    } else {
        const double* p = a.losspart[g] + (e - a.nent[g]);
        for (int wv = b0 * 4; wv < b1 * 4; ++wv) s += p[(size_t)wv * a.K];
    }
    a.tmp[g][(size_t)chunk * (a.nent[g] + a.K) + e] = s;
}
AUX_DEV void reduce2_body(int r, const Reduce2Args& a) {
    if (r < a.P) {
        double s = 0.0;
        for (int i = a.row_ptr[r]; i < a.row_ptr[r + 1]; ++i) {
            const int g = a.row_grp[i];
            if (!a.active[g]) continue;
            const double* t = a.tmp[g] + a.row_ent[i];
            for (int ch = 0; ch < a.nsplit[g]; ++ch) s += t[(size_t)ch * a.stride[g]];
        }
        a.out[r] = (float)s;
    } else if (r < a.P + a.K) {
        const int k = r - a.P;
        double s = 0.0;
        for (int g = 0; g < a.ngroups; ++g) {
            if (!a.active[g]) continue;
            const double* t = a.tmp[g] + a.nent[g] + k;
            for (int ch = 0; ch < a.nsplit[g]; ++ch) s += t[(size_t)ch * a.stride[g]];
        }
        a.out[r] = (float)s;
        if (a.lossraw) a.lossraw[k] = s;
    }
}

#ifdef PINN_EMU
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream) {
    for (int i = 0; i < n; ++i) pack_body(i, packed, idx, theta);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream) {
    for (int j = 0; j < np; ++j) params_body(j, params, theta, defaults, ne, p_off);
}
inline void launch_reduce(const Reduce1Args& a1, const Reduce2Args& a2, int max_n1, int max_split, plat_stream) {
    for (int g = 0; g < a2.ngroups; ++g)
        for (int ch = 0; ch < max_split; ++ch)
            for (int e = 0; e < max_n1; ++e) reduce1_body(e, ch, g, a1);
    for (int r = 0; r < a2.P + a2.K; ++r) reduce2_body(r, a2);
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
__global__ void k_reduce1(const Reduce1Args a) {
    reduce1_body((int)(blockIdx.x * blockDim.x + threadIdx.x), (int)blockIdx.y, (int)blockIdx.z, a);
}
__global__ void k_reduce2(const Reduce2Args a) {
    reduce2_body((int)(blockIdx.x * blockDim.x + threadIdx.x), a);
}
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream st) {
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, st, packed, idx, theta, n);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream st) {
    if (np > 0) hipLaunchKernelGGL(k_params, dim3(1), dim3(64), 0, st, params, theta, defaults, np, ne, p_off);
}
inline void launch_reduce(const Reduce1Args& a1, const Reduce2Args& a2, int max_n1, int max_split, plat_stream st) {
    hipLaunchKernelGGL(k_reduce1, dim3((max_n1 + 255) / 256, max_split, a2.ngroups), dim3(256), 0, st, a1);
    hipLaunchKernelGGL(k_reduce2, dim3((a2.P + a2.K + 255) / 256), dim3(256), 0, st, a2);
}
#endif

}  // namespace aux
