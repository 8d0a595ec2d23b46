// Diagnostic: in-kernel phase clocks (s_memtime) of conv_igemm_bf16x3_ws_kernel on the SPADE mlp_shared shape
// (8 x 512 x 256 x 256 -> 128, 3x3).  Built with -DAG2V_STAMPS; not part of libag2v.so.  GPU box only.
#ifndef NO_STAMPS
#define AG2V_STAMPS 1
#endif
#include "../../ag2video_amd/csrc/conv_bf16x3.hip"
#include "../../ag2video_amd/csrc/conv_halo.hip"      // launch_igemm_bf16x3 hands 3x3 stride-1 split-word launches to these (AG2V_NO_HALO=1 keeps the stamped kernel)
void set_last_conv_kernel(const char*) {}
int launch_act_mask(float*, const float*, long long, int, hipStream_t, int) { return 0; }
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int N = 8, C = argc > 2 ? atoi(argv[2]) : 512, H = 256, W = 256, K = argc > 1 ? atoi(argv[1]) : 128, R = 3;
    const size_t xn = (size_t)N * C * H * W, wn = (size_t)K * C * R * R, yn = (size_t)N * K * H * W;
    float *x, *y; unsigned* at; unsigned long long* st;
    hipMalloc(&x, xn * 4); hipMalloc(&y, yn * 4); hipMalloc(&at, wn * 4);
    std::vector<float> hx(1 << 22);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t o = 0; o < xn; o += hx.size()) hipMemcpy(x + o, hx.data(), std::min(hx.size(), xn - o) * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> hw(wn);
    for (size_t i = 0; i < wn; ++i) hw[i] = 0x3c003800u ^ (unsigned)(i * 40503u);
    hipMemcpy(at, hw.data(), wn * 4, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.a_src = (const float*)at; a.b_src = x; a.out = y;
    a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = R; a.stride = 1; a.pad = 1; a.pad_w = 1; a.OH = H; a.OW = W;
    a.o_mul = 1; a.o_H = H; a.o_W = W;
    a.M = K; a.NG = N * H * W; a.KG = C * R * R; a.a_bytes = (unsigned)(wn * 4); a.b_bytes = (unsigned)(xn * 4);
    const bool tall = a.M >= 192;
    const int tiles = ((a.M + (tall ? 255 : 127)) / (tall ? 256 : 128)) * (a.NG / (tall ? 128 : 256));
    hipMalloc(&st, (size_t)tiles * 16 * 4 * 8); hipMemset(st, 0, (size_t)tiles * 16 * 4 * 8);
#ifdef AG2V_STAMPS
    a.stamps = st;
#endif
    a.tap_major = getenv("TAPK") ? 1 : 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 3;     // REPS=300: sustained clocks instead of a cold burst
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        launch_igemm_bf16x3(a, 0, false);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> hs((size_t)tiles * 16 * 4);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double p[3] = {0, 0, 0}, c[2] = {0, 0}, iters = 0; long np = 0, nc = 0;
    for (int b = 0; b < tiles; ++b)
        for (int w = 0; w < 16; ++w) {
            const unsigned long long* o = &hs[((size_t)b * 16 + w) * 4];
            if (w < 8) { p[0] += o[0]; p[1] += o[1]; p[2] += o[2]; ++np; iters += o[3]; }
            else { c[0] += o[0]; c[1] += o[1]; ++nc; }
        }
    const double it = iters / np;
    if (getenv("PER_WAVE")) {
        for (int w = 0; w < 16; ++w) {
            double a0 = 0, a1 = 0, a2 = 0;
            for (int b = 0; b < tiles; ++b) { const unsigned long long* o = &hs[((size_t)b * 16 + w) * 4]; a0 += o[0]; a1 += o[1]; a2 += o[2]; }
            if (w < 8) printf("  wave %2d (producer, SIMD %d): fetch %.0f stash %.0f barrier %.0f\n", w, w & 3, a0 / tiles / it, a1 / tiles / it, a2 / tiles / it);
            else printf("  wave %2d (consumer, SIMD %d): compute %.0f barrier %.0f\n", w, w & 3, a0 / tiles / it, a1 / tiles / it);
        }
    }
    printf("M=%d %s tile: %.3f ms, %.1f TF-equivalent, %d tiles, %.0f K-tiles per workgroup\n", K, tall ? "256x128" : "128x256", ms,
           2.0 * a.M * a.NG * a.KG / ms / 1e9, tiles, it);
    printf("s_memtime ticks (100 MHz? see below) per K-tile, mean over waves:\n");
    printf("  producer: fetch-issue %.1f  stash(wait+split+lds) %.1f  barrier %.1f  | sum %.1f\n", p[0] / np / it, p[1] / np / it,
           p[2] / np / it, (p[0] + p[1] + p[2]) / np / it);
    printf("  consumer: lds-read+mfma %.1f  barrier %.1f  | sum %.1f\n", c[0] / nc / it, c[1] / nc / it, (c[0] + c[1]) / nc / it);
    printf("  wall per K-tile from the event time: %.1f ns (x workgroups per CU %.2f)\n", ms * 1e6 / it / ((double)tiles / 256), (double)tiles / 256);
    return 0;
}
