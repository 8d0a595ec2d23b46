// Probe: cycles per ds_read_b128 / ds_write_b128 for the fragment access patterns of the bf16x3 kernels as a function of the
// LDS row stride.  One wave per workgroup, one workgroup per CU.  GPU box only; not part of libag2v.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// mode 0: 16x16x32 operand (row = lane & 15, 16-byte chunk = lane >> 4)
// mode 1: 32x32x16 operand (row = lane & 31, chunk = lane >> 5)
// mode 2: producer store (row = lane, chunk fixed)
__global__ __launch_bounds__(256) void probe(unsigned* out, int stride_bytes, int mode, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    int row, chunk;
    if (mode == 0) { row = lane & 15; chunk = lane >> 4; }
    else if (mode == 1) { row = lane & 31; chunk = lane >> 5; }
    else { row = lane; chunk = 0; }
    const unsigned addr = row * stride_bytes + chunk * 16;
    const unsigned blk = (mode == 2 ? 0 : 16) * stride_bytes;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<volatile u32x4*>(lds + addr + (u & 3) * 16) = acc;
        } else {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<volatile u32x4*>(lds + addr + (u & 3) * blk + (u >> 2) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (unsigned)(t1 - t0); }
    out[1 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
    const int iters = 2000;
    const int strides[] = {64, 80, 96, 112, 144, 160, 176, 208, 272};
    for (int mode = 0; mode < 3; ++mode) {
        printf("mode %d (%s):", mode, mode == 0 ? "16x16x32 read" : mode == 1 ? "32x32x16 read" : "row-per-lane store");
        for (int s : strides) {
            if (64 * s + 256 > 65536) { printf("  %d: -", s); continue; }
            unsigned h = 0;
            probe<<<1, 256, 65536>>>(d, s, mode, iters);
            probe<<<1, 256, 65536>>>(d, s, mode, iters);
            hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("  %d: %.1f", s, (double)h / (iters * 8 * 4));
        }
        printf("   (s_memtime ticks per wave instruction, 4 waves issuing)\n");
    }
    return 0;
}
