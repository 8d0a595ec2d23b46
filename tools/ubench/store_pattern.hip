// Probe: HBM write throughput of the layout convolution's store pattern against a contiguous fill, 120 x 512 x 256 x 256 floats (16.1 GB).
//   P0  contiguous: workgroup b writes 64 KB at b * 64 KB
//   P1  layout_conv_fwd_kernel's: workgroup = (image, 4 rows), loops over 32 blocks of 16 channels; wave w stores row oy + w (1 KB) of each channel
//   P2  the same workgroups, wave w stores all 4 rows (4 KB contiguous) of channels k0 + 4 w .. + 3
//   P3  P1 with the channel blocks walked by gridDim.z = 32 workgroups instead of a loop (more, shorter workgroups)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void p0(float4* out) {
    float4* o = out + (size_t)blockIdx.x * 4096;
    for (int i = threadIdx.x; i < 4096; i += 256) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void p1(float* out, int K, int H, int W, int kb0, int kbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, oy = blockIdx.x * 4 + wave, n = blockIdx.y;
    for (int k0 = (kb0 + blockIdx.z) * 16; k0 < K; k0 += kbs * 16)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            *reinterpret_cast<float4*>(out + (((size_t)n * K + k0 + kk) * H + oy) * W + lane * 4) = make_float4(1.f, 2.f, 3.f, (float)kk);
}
__global__ __launch_bounds__(256) void p2(float* out, int K, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, oy = blockIdx.x * 4, n = blockIdx.y;
    for (int k0 = 0; k0 < K; k0 += 16)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(out + (((size_t)n * K + k0 + wave * 4 + kk) * H + oy + r) * W + lane * 4) = make_float4(1.f, 2.f, 3.f, (float)kk);
}
int main() {
    const int NB = 120, K = 512, H = 256, W = 256;
    const size_t n = (size_t)NB * K * H * W;
    float* out; hipMalloc(&out, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int p = 0; p < 4; ++p) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (p == 0) hipLaunchKernelGGL(p0, dim3((unsigned)(n / 16384)), dim3(256), 0, 0, (float4*)out);
            if (p == 1) hipLaunchKernelGGL(p1, dim3(H / 4, NB, 1), dim3(256), 0, 0, out, K, H, W, 0, 1);
            if (p == 2) hipLaunchKernelGGL(p2, dim3(H / 4, NB), dim3(256), 0, 0, out, K, H, W);
            if (p == 3) hipLaunchKernelGGL(p1, dim3(H / 4, NB, 32), dim3(256), 0, 0, out, K, H, W, 0, 32);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("P%d: %.2f ms = %.2f TB/s\n", p, best, n * 4 / best / 1e9);
    }
    return 0;
}
