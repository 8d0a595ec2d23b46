// Micro-benchmarks backing DESIGN.md's load-path choices (GPU box only):
//  1. raw buffer_load_dwordx4 range checking: per dword or per access?  (offset -4 and bytes-12)
//  2. wave-instruction issue rate of coalesced dword vs dwordx4 buffer loads from an L2-resident array.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__global__ void range_kernel(const unsigned* src, unsigned bytes, unsigned* out) {
    auto r = rsrc(src, bytes);
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, 0xFFFFFFFCu, 0, 0);   // starts 4 B before the buffer
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, bytes - 12u, 0, 0);   // last dword is past the end
    u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(r, 4u, 0, 0);            // misaligned, in range
    for (int i = 0; i < 4; ++i) { out[i] = a[i]; out[4 + i] = b[i]; out[8 + i] = c[i]; }
}
template <int VEC>
__global__ __launch_bounds__(256) void rate_kernel(const unsigned* src, unsigned bytes, unsigned* out, int iters, int misalign) {
    auto r = rsrc(src, bytes);
    const unsigned lane_off = (threadIdx.x * VEC + misalign) * 4u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned off = lane_off + ((it * 8 + u) & 63) * 4096u;   // 256 KB window, L2 resident
            if (VEC == 4) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
                acc += v[0] ^ v[1] ^ v[2] ^ v[3];
            } else {
                acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const unsigned n = 1 << 20;
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = i + 1;
    unsigned *d, *o;
    hipMalloc(&d, n * 4 + 4096); hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    range_kernel<<<1, 1>>>(d, n * 4, o);
    unsigned ho[12];
    hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
    printf("load at -4      : %u %u %u %u   (per-dword check => 0 1 2 3)\n", ho[0], ho[1], ho[2], ho[3]);
    printf("load at bytes-12: %u %u %u %u   (per-dword check => n-2 n-1 n 0; n=%u)\n", ho[4], ho[5], ho[6], ho[7], n);
    printf("load at +4      : %u %u %u %u   (=> 2 3 4 5)\n", ho[8], ho[9], ho[10], ho[11]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 2;
    for (int mis = 0; mis < 2; ++mis)
        for (int vec = 1; vec <= 4; vec += 3) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (vec == 4) rate_kernel<4><<<blocks, 256>>>(d, n * 4, o, iters, mis);
                else rate_kernel<1><<<blocks, 256>>>(d, n * 4, o, iters, mis);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double winst = (double)blocks * 4 * iters * 8;          // wave-level load instructions
            const double per_cu = winst / 256;
            printf("vec=%d misalign=%d: %.3f ms, %.1f ns per wave-load per CU (%.1f cycles @2.4GHz), %.0f GB/s\n", vec, mis, ms,
                   ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4, winst * 64 * vec * 4 / ms / 1e6);
        }
    return 0;
}
