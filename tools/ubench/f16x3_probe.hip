// Accuracy probe: fp32 dot products on the 16-bit matrix pipe with 3-term splits, bf16 (x = hi + lo, 8 + 8 bits) against fp16
// (11 + 11 bits).  One wave computes a 32 x 32 x K product with v_mfma_f32_32x32x16_{bf16,f16}; reference in fp64 on the host.
// Also answers whether the MFMA keeps fp16 SUBNORMAL inputs (the lo term of a small value is a subnormal fp16).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/f16x3_probe.hip -o tools/ubench/f16x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool F16>
__global__ void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    // A [32][K], B [32][K] row-major; C[i][j] = sum_k A[i][k] B[j][k]
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        for (int e = 0; e < 8; ++e) { a[e] = A[r * K + k0 + 8 * h + e]; b[e] = B[r * K + k0 + 8 * h + e]; }
        if (F16) {
            f16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                ah[e] = (_Float16)a[e]; al[e] = (_Float16)(a[e] - (float)ah[e]);
                bh[e] = (_Float16)b[e]; bl[e] = (_Float16)(b[e] - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        } else {
            bf16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; ++e) {
                ah[e] = (__bf16)a[e]; al[e] = (__bf16)(a[e] - (float)ah[e]);
                bh[e] = (__bf16)b[e]; bl[e] = (__bf16)(b[e] - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = acc[i];
    }
}

int main() {
    const int K = 4608;
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float scalesA[] = {1.f, 1e-1f, 1e-2f, 1e-3f, 1e-4f};
    const float scalesB[] = {1.f, 3e-2f, 3e-2f, 3e-2f, 1e-4f};
    float *dA, *dB, *dC;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC, 32 * 32 * 4);
    for (int t = 0; t < 5; ++t) {
        std::vector<float> A(32 * K), B(32 * K), C(1024);
        for (auto& v : A) { v = nd(gen) * scalesA[t]; if (v < 0) v *= 0.2f; }      // LeakyReLU-shaped activations
        for (auto& v : B) v = nd(gen) * scalesB[t];
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        double ref[1024], rms = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k];
            ref[i * 32 + j] = s; rms += s * s;
        }
        rms = std::sqrt(rms / 1024);
        double err[2];
        for (int m = 0; m < 2; ++m) {
            if (m) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double e = 0; for (int i = 0; i < 1024; ++i) e = std::max(e, std::fabs(C[i] - ref[i]));
            err[m] = e / rms;
        }
        // fp32 sequential reference error for scale
        double e32 = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s); e32 = std::max(e32, std::fabs(s - ref[i * 32 + j])); }
        printf("scale A %.0e B %.0e: max err / rms(out): bf16x3 %.3e   f16x3 %.3e   (fp32 fmaf chain %.3e)\n", scalesA[t], scalesB[t], err[0], err[1], e32 / rms);
    }
    return 0;
}
