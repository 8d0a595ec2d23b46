// Shared helpers for the ag2v gfx950 kernels (MI355X / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AG2V_OK 0
#define AG2V_EINVAL (-1)
#define AG2V_EHIP (-2)

#define AG2V_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return AG2V_EHIP;              \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for 256-thread blocks (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sh4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// NV block-wide sums of a 256-thread block at once: the wave sums go to LDS together and ONE barrier hands them over (block_sum_256
// costs two barriers per value; a reduce kernel with 4-8 sums per 4096-element chunk spent its time in them).  sh: 4 * NV floats, not
// reused before the next barrier of the caller.  Same summation order as block_sum_256: bit-identical results.
template <int NV>
__device__ __forceinline__ void block_sums_256(float (&v)[NV], float* sh) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[(threadIdx.x >> 6) * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = sh[k] + sh[NV + k] + sh[2 * NV + k] + sh[3 * NV + k];
}

// Bijective XCD-aware remap (8 XCDs, blocks are dealt round-robin): gives every XCD a
// contiguous range of logical tile ids so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
