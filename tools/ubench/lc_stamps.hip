// Diagnostic: phase clocks (s_memtime) of layout_conv_fwd_kernel<3, 16> on conv_dim_in's c3 shape (22 slots, 6 occupied, 512 channels,
// 120 x 256 x 256), one workgroup's first 8 channel blocks, all four waves.  Built with -DAG2V_LC_STAMPS; GPU box only:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/ubench/lc_stamps.hip -o tools/ubench/lc_stamps && tools/ubench/lc_stamps [img]
#define AG2V_LC_STAMPS 1
#include "../../ag2video_amd/csrc/layout_warp.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int NB = 120, J = 22, K = 512, H = 256, W = 256, R = 3;
    const bool with_img = argc > 1;
    std::vector<float> hU((size_t)NB * J * K * 9), hb((size_t)NB * J * 4, 0.f), himg((size_t)NB * 3 * H * W, 0.5f), hw((size_t)K * 27, 0.1f);
    std::vector<int> hk((size_t)NB * J, 1);
    for (size_t i = 0; i < hU.size(); ++i) hU[i] = (float)((i * 2654435761u) % 2001) / 10000.f - 0.1f;
    const int occ[6] = {0, 1, 2, 11, 12, 13};
    for (int n = 0; n < NB; ++n)
        for (int o = 0; o < 6; ++o) {
            float* b = &hb[((size_t)n * J + occ[o]) * 4];
            b[0] = 0.05f + 0.07f * ((n + 3 * o) % 7); b[1] = 0.05f + 0.06f * ((n + 5 * o) % 8); b[2] = 0.1f + 0.03f * ((n + o) % 9); b[3] = 0.1f + 0.03f * ((n + 2 * o) % 9);
        }
    float *U, *b, *out, *img, *w; int* keep;
    hipMalloc(&U, hU.size() * 4); hipMalloc(&b, hb.size() * 4); hipMalloc(&keep, hk.size() * 4); hipMalloc(&out, (size_t)NB * K * H * W * 4);
    hipMalloc(&img, himg.size() * 4); hipMalloc(&w, hw.size() * 4);
    hipMemcpy(U, hU.data(), hU.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(keep, hk.data(), hk.size() * 4, hipMemcpyHostToDevice); hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        ag2v_layout_conv_img_fwd(U, b, keep, nullptr, out, NB, J, K, H, W, R, 1, 1, 0, with_img ? img : nullptr, with_img ? w : nullptr, nullptr, with_img ? 3 : 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("launch %.3f ms (%s)\n", ms, with_img ? "with image channels" : "layout term only");
    unsigned long long st[4][8][6];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lc_stamps), sizeof(st));
    printf("clock ticks (s_memtime) per channel block: top barrier -> taps built | -> second barrier passed | -> accumulators initialised (+ image channels) | -> object loop done | -> stores issued | -> next block's top\n");
    for (int wv = 0; wv < 4; ++wv)
        for (int k = 0; k < 7; ++k)
            printf("wave %d block %d: %6lld %6lld %6lld %6lld %6lld %6lld\n", wv, k, (long long)(st[wv][k][1] - st[wv][k][0]), (long long)(st[wv][k][2] - st[wv][k][1]),
                   (long long)(st[wv][k][4] - st[wv][k][2]), (long long)(st[wv][k][5] - st[wv][k][4]), (long long)(st[wv][k][3] - st[wv][k][5]),
                   (long long)(st[wv][k + 1][0] - st[wv][k][3]));
    return 0;
}
