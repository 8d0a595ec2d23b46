// Probe: the per-CU issue rate of 1-KiB wave instructions on the vector-memory path -- LDS-DMA loads (buffer_load_dwordx4 ... lds), the
// same with every lane out of range (the "dummy" pieces of conv_halo.hip), and global_store_dwordx4 -- with 1, 2, 4 or 8 waves of a
// 512-thread workgroup issuing, one workgroup per CU, the source resident in L2.  conv_halo.hip moves 896 + 128 such instructions per
// 128-channel tile: DESIGN.md 13.3 prices the kernel against this rate.  GPU box only:  hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate_probe tools/dma_rate_probe.hip && /tmp/dma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_vptr;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// mode 0: LDS-DMA loads, 1: LDS-DMA with every lane out of range, 2: 16-byte global stores (contiguous 1 KiB per instruction),
// 3: 16-byte global stores, 16 rows x 64 bytes per instruction (the halo kernels' epilogue pattern)
__global__ __launch_bounds__(512) void probe(const float* src, float* dst, long long* cycles, int iters, int waves, int mode) {
    __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* base = src + (size_t)blockIdx.x * 16384;                    // 64 KiB per workgroup
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 65536, 0x00020000);
    float* out = dst + (size_t)blockIdx.x * (1 << 20) + (size_t)wave * (1 << 17);      // 4 MiB per workgroup, 512 KiB per wave
    const float4 val = make_float4(1.f, 2.f, 3.f, (float)lane);
    __syncthreads();
    const long long t0 = clock64();
    if (wave < waves) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = i * 8 + u;
                if (mode <= 1) {
                    const unsigned voff = mode == 1 ? 0xFFFFFFF0u : (unsigned)(((k * 8 + wave) & 63) * 1024 + lane * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(lds + ((k & 15) * 8 + wave) * 1024), 16, voff, 0, 0, 0);
                } else if (mode == 2) {
                    *reinterpret_cast<float4*>(out + (size_t)((k & 127) * 256) + lane * 4) = val;
                } else {
                    *reinterpret_cast<float4*>(out + (size_t)(lane & 15) * 8192 + (size_t)((k & 127) * 16) + (lane >> 4) * 4) = val;
                }
            }
            wait_vm<8>();
        }
        wait_vm<0>();
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    const int nb = 256;
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, (size_t)nb * 65536);
    hipMalloc(&dst, (size_t)nb * (4 << 20));
    hipMalloc(&cyc, nb * sizeof(long long));
    hipMemset(src, 0, (size_t)nb * 65536);
    long long h[256];
    const char* names[4] = {"LDS-DMA load, 1 KiB", "LDS-DMA, all lanes out of range", "store 16 B/lane, 1 KiB contiguous", "store 16 B/lane, 16 rows x 64 B"};
    for (int mode = 0; mode < 4; ++mode)
        for (int waves = 1; waves <= 8; waves *= 2) {
            const int iters = 512;
            probe<<<nb, 512>>>(src, dst, cyc, 8, waves, mode);               // warm
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            probe<<<nb, 512>>>(src, dst, cyc, iters, waves, mode);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            (void)h;
            const double ns = (double)ms * 1e6 / ((double)iters * 8 * waves);           // per instruction and CU (every CU runs one workgroup)
            printf("%-36s %d waves issuing: %7.1f ns per instruction and CU  (%5.1f B/ns/CU, %5.2f TB/s over 256 CUs)\n", names[mode], waves, ns,
                   1024.0 / ns, 1024.0 / ns * 256 / 1000.0);
        }
    return 0;
}
