// Layout render (box splat), optical-flow warp and the small resampling kernels of the AG2Video hot path.
//
//  * layout render: closed form of boxes_to_layout (models/layout.py:28-63,98-130,205-237): every object's D-vector is
//    painted into its box by bilinear sampling of a constant 8x8 image (align_corners=True, zero padding) and objects are
//    summed:  out[d,y,x] = sum_o v[o,d] * tau(7 (X_x - x0_o)/w_o) * tau(7 (Y_y - y0_o)/h_o),  tau(i) = clamp(min(i+1, 8-i), 0, 1).
//    The reference materialises [n_obj, D, H, W]; here one pass writes [D, H, W] once (write-bandwidth bound) and the
//    backward reads only each object's support rectangle.
//  * flow warp: resample/get_grid (models/utils.py:113-140): grid_sample(bilinear, border, align_corners=False).
//  * avg_pool2d(3, s2, p1, count_include_pad=False) (discriminator.py:270-271), nearest x2 upsample
//    (spade_generator.py:36,55-70; flows_generator.py:47), nearest resize of the segmap (normalization.py:102).
#include <algorithm>
#include "common.h"

#define MAXO 16

__device__ __forceinline__ float lin01(int i, int steps) {
    // torch.linspace(0, 1, steps)[i]: symmetric evaluation (start + i*step | end - (steps-1-i)*step)
    if (steps == 1) return 0.f;
    const float step = 1.0f / (float)(steps - 1);
    return (i < steps / 2) ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}
__device__ __forceinline__ float tau8(float pos, float lo, float extent) {
    // grid = ((P - lo)/extent)*2 - 1 ; sample position in the 8-px image (align_corners=True): ((g+1)/2)*7
    const float g = ((pos - lo) / extent) * 2.0f - 1.0f;
    const float i = ((g + 1.0f) * 0.5f) * 7.0f;
    return fminf(fmaxf(fminf(i + 1.0f, 8.0f - i), 0.0f), 1.0f);
}

// grid: (ceil(H/4), BT, DSPLIT); 4 waves, wave w renders row y = 4*blockIdx.x + w for the block's channel range.
__global__ __launch_bounds__(256) void layout_render_fwd_kernel(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                                const int* __restrict__ keep, float* __restrict__ out, int O,
                                                                int D, int H, int W, int dper) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = blockIdx.x * 4 + wave;
    const int bt = blockIdx.y;
    const int d0 = blockIdx.z * dper, d1 = min(D, d0 + dper);
    if (y >= H) return;
    const float Y = lin01(y, H);
    float tyv[MAXO];
    float bx0[MAXO], bw[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        tyv[o] = 0.f; bx0[o] = 0.f; bw[o] = 1.f;
        if (o < O) {
            const float* b = boxes + ((size_t)bt * O + o) * 4;
            const float x0 = b[0], y0 = b[1], ww = b[2], hh = b[3];
            const bool legal = (x0 != 0.f) || (y0 != 0.f) || (ww != 0.f) || (hh != 0.f);
            if (legal && keep[(size_t)bt * O + o] != 0) {
                tyv[o] = tau8(Y, y0, hh);
                bx0[o] = x0; bw[o] = ww;
            }
        }
    }
    const int W4 = W >> 2;
    for (int xq = lane; xq < W4; xq += 64) {
        float tx[MAXO][4];
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) tx[o][j] = (tyv[o] != 0.f) ? tau8(lin01(xq * 4 + j, W), bx0[o], bw[o]) * tyv[o] : 0.f;
        for (int d = d0; d < d1; ++d) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int o = 0; o < MAXO; ++o) {
                if (tyv[o] != 0.f) {  // wave-uniform
                    const float v = vecs[((size_t)bt * O + o) * D + d];
                    a0 += v * tx[o][0]; a1 += v * tx[o][1]; a2 += v * tx[o][2]; a3 += v * tx[o][3];
                }
            }
            *reinterpret_cast<float4*>(out + (((size_t)bt * D + d) * H + y) * W + xq * 4) = make_float4(a0, a1, a2, a3);
        }
    }
}

// grid: (O, BT, ceil(D/32)); wave w reduces channels d = 32*blockIdx.z + w + 4*i over the object's support rectangle.
__global__ __launch_bounds__(256) void layout_render_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                                const int* __restrict__ keep, float* __restrict__ dvecs, int O,
                                                                int D, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x, bt = blockIdx.y;
    const float* b = boxes + ((size_t)bt * O + o) * 4;
    const float x0 = b[0], y0 = b[1], ww = b[2], hh = b[3];
    const bool legal = (x0 != 0.f) || (y0 != 0.f) || (ww != 0.f) || (hh != 0.f);
    const bool active = legal && keep[(size_t)bt * O + o] != 0;
    for (int i = 0; i < 8; ++i) {
        const int d = blockIdx.z * 32 + wave + 4 * i;
        if (d >= D) break;
        float acc = 0.f;
        if (active) {
            for (int xb = 0; xb < W; xb += 64) {
                const int x = xb + lane;
                const float txv = (x < W) ? tau8(lin01(x, W), x0, ww) : 0.f;
                if (__ballot(txv != 0.f) == 0ull) continue;  // wave-uniform skip of empty column blocks
                const float* dp = dout + ((size_t)bt * D + d) * H * W + x;
                for (int y = 0; y < H; ++y) {
                    const float tyv = tau8(lin01(y, H), y0, hh);  // wave-uniform
                    if (tyv != 0.f && txv != 0.f) acc += dp[(size_t)y * W] * (txv * tyv);
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) dvecs[((size_t)bt * O + o) * D + d] = acc;
    }
}

// ---------------------------------------------------------------- layout convolution (low-rank form)
// A layout is rank-<=n_obj:  seg[d,y,x] = sum_j v[j,d] * w_j(y,x),  w_j = ty_j (x) tx_j.  A convolution of it,
//   Y[k,oy,ox] = sum_{d,r,s} Wt[k,d,r,s] * seg[d, oy*st+r-pad, ox*st+s-pad],
// therefore equals  sum_j sum_{r,s} U[j,k,r,s] * w_j(oy*st+r-pad, ox*st+s-pad)  with  U[j,k,r,s] = sum_d Wt[k,d,r,s] v[j,d]
// (a tiny GEMM done by the caller).  This kernel evaluates that sum directly -- the [D,H,W] layout (1024 x 256 x 256 floats
// per sample and frame pair in the generator) is never materialised and ~99 % of the dense convolution's FLOPs disappear
// ("layout tensors are huge and low-rank").  Reference call sites: the seg channels of
// generator.py:74,82-83 (conv_dim_in) and flows_generator.py:32 (down_flow[0]).  Zero padding: w_j is 0 outside [0,H)x[0,W).
// U: [NB, J, K, R*R]; boxes [NB, J, 4]; keep [NB, J]; residual/out [NB, K, OH, OW].
// 1-D box profile at index i of the map the convolution sees.  pool == 0: the full-resolution profile tau8(lin01(i, L)).
// pool == 1: the map is avg_pool2d(3, stride 2, pad 1, count_include_pad=False) of the layout (the discriminator's second
// scale, discriminator.py:270-271,351); pooling a separable map is separable, so the pooled profile is the average of the
// valid taps 2i-1..2i+1 of the full-resolution profile (L = full-resolution length).
__device__ __forceinline__ float lc_profile(int i, int L, int pool, float lo, float ext) {
    if (!pool) return tau8(lin01(i, L), lo, ext);
    float s = 0.f;
    int c = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int k = 2 * i - 1 + d;
        if ((unsigned)k < (unsigned)L) { s += tau8(lin01(k, L), lo, ext); ++c; }
    }
    return s / (float)c;
}

template <int R>
__device__ __forceinline__ void lc_taps_y(float (&tyr)[R], int oy, int stride, int pad, int Hm, int H, int pool, float y0, float hh,
                                          bool active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yy = oy * stride + r - pad;
        tyr[r] = (active && (unsigned)yy < (unsigned)Hm) ? lc_profile(yy, H, pool, y0, hh) : 0.f;
    }
}

constexpr int LC_TJ = 24, LC_TW = 264;   // table limits of layout_conv_fwd_kernel

// img / wimg / iscale (nullable; stride 1, pool 0): the IMAGE channels of the same convolution -- conv_dim_in and down_flow[0] see
// cat[layouts, frame(s)] (generator.py:66,74,82) -- are added here: out += iscale[n] * conv(img[n, :IC], wimg[K, IC, R, R]).  Their
// IC * R * R = 27 multiply-adds per output are nothing beside the 16 GB this kernel writes at c3, and the dense launch, its 16 GB
// output and the read-back of it as `residual` disappear (5.9 + ~3 ms per step).
template <int R, int KPER>
__global__ __launch_bounds__(256) void layout_conv_fwd_generic_kernel(const float* __restrict__ U, const float* __restrict__ boxes,
                                                              const int* __restrict__ keep, const float* __restrict__ residual,
                                                              float* __restrict__ out, int J, int K, int H, int W, int Hm, int Wm,
                                                              int pool, int OH, int OW, int stride, int pad,
                                                              const float* __restrict__ img, const float* __restrict__ wimg,
                                                              const float* __restrict__ iscale, int IC) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oy = blockIdx.x * 4 + wave;
    const int n = blockIdx.y;
    const int k0 = blockIdx.z * KPER;
    // column profiles tx_j(x) of every object of this image, evaluated once per workgroup (the pooled profile costs three
    // tau8 evaluations per value and every output pixel needs R of them per object)
    constexpr int TJ = 16, TW = 264;
    __shared__ float txp[TJ][TW];
    const bool tabled = (J <= TJ) && (Wm <= TW);
    if (tabled) {
        for (int idx = threadIdx.x; idx < J * Wm; idx += 256) {
            const int j = idx / Wm, xx = idx - j * Wm;
            const float* b = boxes + ((size_t)n * J + j) * 4;
            txp[j][xx] = lc_profile(xx, W, pool, b[0], b[2]);
        }
    }
    __syncthreads();
    if (oy >= OH) return;
    // row taps ty_j(oy * stride + r - pad) of every object, one (j, r) per lane, read back with readlane in the object loop:
    // evaluated redundantly by all 64 lanes for all J objects they cost more than the FMAs of the 2-3 objects a row touches
    const bool lane_taps = J * R <= 64;
    float my_ty = 0.f;
    if (lane_taps && lane < J * R) {
        const int jj = lane / R, rr = lane - jj * R;
        const float* b = boxes + ((size_t)n * J + jj) * 4;
        const float bx = b[0], by = b[1], bw = b[2], bh = b[3];
        const bool act = ((bx != 0.f) || (by != 0.f) || (bw != 0.f) || (bh != 0.f)) && keep[(size_t)n * J + jj] != 0;
        const int yy = oy * stride + rr - pad;
        my_ty = (act && (unsigned)yy < (unsigned)Hm) ? lc_profile(yy, H, pool, by, bh) : 0.f;
    }
    const int nq = (OW + 3) >> 2;
    // pixel q of thread-iteration xq: 4 consecutive pixels (one 16-byte access) when rows are 16-byte aligned, else the
    // lane-contiguous pixels xq_block * 256 + lane + 64 q (coalesced 4-byte accesses; OW = 129 in the discriminator)
    const bool vec = (OW & 3) == 0;
    auto px = [&](int xq, int q) -> int { return vec ? xq * 4 + q : (xq >> 6) * 256 + (xq & 63) + 64 * q; };
    for (int xq = lane; vec ? (xq < nq) : ((xq >> 6) * 256 < OW); xq += 64) {
        float acc[KPER][4];
#pragma unroll
        for (int kk = 0; kk < KPER; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ox = px(xq, q);
                acc[kk][q] = (residual && (k0 + kk) < K && ox < OW)
                                 ? residual[(((size_t)n * K + k0 + kk) * OH + oy) * OW + ox] : 0.f;
            }
        if (img) {      // dense image channels (stride 1, no pooling; vec layout: 4 consecutive pixels per thread)
            const float sc = iscale ? iscale[n] : 1.f;
            const int x0i = xq * 4 - pad;
            for (int c = 0; c < IC; ++c) {
                const float* ip = img + ((size_t)n * IC + c) * H * W;
                float iv[R][R + 3];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int yy = oy + r - pad;
                    const bool rowok = (unsigned)yy < (unsigned)H;
#pragma unroll
                    for (int i = 0; i < R + 3; ++i) {
                        const int xx = x0i + i;
                        iv[r][i] = (rowok && (unsigned)xx < (unsigned)W) ? ip[yy * W + xx] : 0.f;
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KPER; ++kk) {
                    if (k0 + kk < K) {
                        const float* wk = wimg + (((size_t)(k0 + kk) * IC + c) * R) * R;      // wave-uniform: scalar loads
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int s2 = 0; s2 < R; ++s2) {
                                const float wv = wk[r * R + s2] * sc;
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(wv, iv[r][s2 + q], acc[kk][q]);
                            }
                    }
                }
            }
        }
        for (int j = 0; j < J; ++j) {
            const float* b = boxes + ((size_t)n * J + j) * 4;
            const float x0 = b[0], y0 = b[1], ww = b[2], hh = b[3];
            const bool legal = (x0 != 0.f) || (y0 != 0.f) || (ww != 0.f) || (hh != 0.f);
            const bool active = legal && keep[(size_t)n * J + j] != 0;
            float tyr[R];
            if (lane_taps) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    tyr[r] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_ty), j * R + r));
            } else {
                lc_taps_y<R>(tyr, oy, stride, pad, Hm, H, pool, y0, hh, active);
            }
            float tysum = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) tysum += fabsf(tyr[r]);
            if (tysum == 0.f) continue;  // wave-uniform
            float txs[R][4];
            float txsum = 0.f;
#pragma unroll
            for (int s = 0; s < R; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = px(xq, q) * stride + s - pad;
                    txs[s][q] = ((unsigned)xx < (unsigned)Wm) ? (tabled ? txp[j][xx] : lc_profile(xx, W, pool, x0, ww)) : 0.f;
                    txsum += fabsf(txs[s][q]);
                }
            if (__ballot(txsum != 0.f) == 0ull) continue;
            const float* uj = U + (((size_t)n * J + j) * K + k0) * (R * R);
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk) {
                if (k0 + kk < K) {
                    const float* u = uj + kk * (R * R);
#pragma unroll
                    for (int s = 0; s < R; ++s) {
                        float cs = 0.f;
#pragma unroll
                        for (int r = 0; r < R; ++r) cs += tyr[r] * u[r * R + s];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[kk][q] += cs * txs[s][q];
                    }
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < KPER; ++kk) {
            if (k0 + kk >= K) continue;
            float* orow = out + (((size_t)n * K + k0 + kk) * OH + oy) * OW;
            if (vec) {
                *reinterpret_cast<float4*>(orow + xq * 4) = make_float4(acc[kk][0], acc[kk][1], acc[kk][2], acc[kk][3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (px(xq, q) < OW) orow[px(xq, q)] = acc[kk][q];
            }
        }
    }
}

#ifdef AG2V_LC_STAMPS      // diagnostic build only (tools/ubench/lc_stamps.hip): phase clocks of one workgroup's first channel blocks
__device__ unsigned long long g_lc_stamps[4][8][6];
#define LC_STAMP(slot) do { if (blockIdx.x == 17 && blockIdx.y == 3 && lane == 0 && kbi < 8) g_lc_stamps[wave][kbi][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LC_STAMP(slot) do { } while (0)
#endif
// The kernel the path uses (J <= 24 objects, map rows <= 264 wide; everything else: layout_conv_fwd_generic_kernel below).
// Workgroup = 4 waves = 4 output rows x KPER channels of one image; three LDS tables replace per-lane work that is uniform
// over a wave: (1) the column profiles tx_j(x) of all objects, (2) the R row taps ty_j of each wave's row, (3) per wave and
// (object j, channel kk) the row-contracted taps cs[s] = sum_r ty_j[r] * U[j, k, r, s].  The inner loop is then R x 4 FMAs per
// (object, channel) on 4 pixels, fed by one broadcast ds_read_b128; before, every lane recomputed the 9-FMA contraction (and,
// from 22 objects on, both profiles) itself: 10.8 -> see DESIGN.md ms per launch at c3 (22 objects, 512 channels, 120 x 256 x 256).
template <int R, int KPER>
__global__ __launch_bounds__(256) void layout_conv_fwd_kernel(const float* __restrict__ U, const float* __restrict__ boxes,
                                                              const int* __restrict__ keep, const float* __restrict__ residual,
                                                              float* __restrict__ out, int J, int K, int H, int W, int Hm, int Wm,
                                                              int pool, int OH, int OW, int stride, int pad,
                                                              const float* __restrict__ img, const float* __restrict__ wimg,
                                                              const float* __restrict__ iscale, int IC) {
    constexpr int TJ = LC_TJ, TW = LC_TW;
    __shared__ float txp[TJ][TW];
    __shared__ float tyl[4][TJ][4];
    __shared__ float4 csl[4][TJ][KPER];
    constexpr int RRP = (R * R + 3) & ~3;       // filter taps per (channel, image channel), padded to 16-byte reads
    __shared__ float wl[KPER][8][RRP];          // image filter of this channel block, scaled (IC <= 8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oy = blockIdx.x * 4 + wave;
    const int n = blockIdx.y;
    for (int idx = threadIdx.x; idx < J * Wm; idx += 256) {
        const int j = idx / Wm, xx = idx - j * Wm;
        const float* b = boxes + ((size_t)n * J + j) * 4;
        txp[j][xx] = lc_profile(xx, W, pool, b[0], b[2]);
    }
    for (int idx = lane; idx < J * 4; idx += 64) {
        const int j = idx >> 2, r = idx & 3;
        const float* b = boxes + ((size_t)n * J + j) * 4;
        const float bx = b[0], by = b[1], bw = b[2], bh = b[3];
        const bool act = ((bx != 0.f) || (by != 0.f) || (bw != 0.f) || (bh != 0.f)) && keep[(size_t)n * J + j] != 0;
        const int yy = oy * stride + r - pad;
        tyl[wave][j][r] = (r < R && oy < OH && act && (unsigned)yy < (unsigned)Hm) ? lc_profile(yy, H, pool, by, bh) : 0.f;
    }
    // the profiles above are shared by every channel: the workgroup walks the channel blocks kb = z, z + gridDim.z, ... itself
    int kbi = -1;
    (void)kbi;
    for (int k0 = blockIdx.z * KPER; k0 < K; k0 += gridDim.z * KPER) {
    ++kbi;
    __syncthreads();          // (first pass: the profile tables are complete; later passes: the previous block's tables are no longer read)
    LC_STAMP(0);
    if (img) {
        const float sc = iscale ? iscale[n] : 1.f;
        for (int idx = threadIdx.x; idx < KPER * IC * RRP; idx += 256) {
            const int t = idx % RRP, c = (idx / RRP) % IC, kk = idx / (RRP * IC);
            wl[kk][c][t] = (t < R * R && k0 + kk < K) ? wimg[(((size_t)(k0 + kk) * IC + c) * R) * R + t] * sc : 0.f;
        }
    }
    for (int idx = lane; idx < J * KPER; idx += 64) {
        const int j = idx / KPER, kk = idx - j * KPER;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        const float4 t4 = *reinterpret_cast<const float4*>(&tyl[wave][j][0]);
        const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
        if (k0 + kk < K && (t4.x != 0.f || t4.y != 0.f || t4.z != 0.f || t4.w != 0.f)) {      // rows that miss the object read no taps
            const float* u = U + (((size_t)n * J + j) * K + k0 + kk) * (R * R);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int s2 = 0; s2 < R; ++s2) cs[s2] = fmaf(tv[r], u[r * R + s2], cs[s2]);
        }
        csl[wave][j][kk] = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
    LC_STAMP(1);
    __syncthreads();
    LC_STAMP(2);
    const int nq = (OW + 3) >> 2;
    const bool vec = (OW & 3) == 0;
    auto px = [&](int xq, int q) -> int { return vec ? xq * 4 + q : (xq >> 6) * 256 + (xq & 63) + 64 * q; };
    if (oy < OH)
    for (int xq = lane; vec ? (xq < nq) : ((xq >> 6) * 256 < OW); xq += 64) {
        float acc[KPER][4];
#pragma unroll
        for (int kk = 0; kk < KPER; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ox = px(xq, q);
                acc[kk][q] = (residual && (k0 + kk) < K && ox < OW)
                                 ? residual[(((size_t)n * K + k0 + kk) * OH + oy) * OW + ox] : 0.f;
            }
        if (img) {      // dense image channels (stride 1, no pooling; vec layout: 4 consecutive pixels per thread)
            const int x0i = xq * 4 - pad;
            for (int c = 0; c < IC; ++c) {
                const float* ip = img + ((size_t)n * IC + c) * H * W;
                float iv[R][R + 3];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int yy = oy + r - pad;
                    const bool rowok = (unsigned)yy < (unsigned)H;
#pragma unroll
                    for (int i = 0; i < R + 3; ++i) {
                        const int xx = x0i + i;
                        iv[r][i] = (rowok && (unsigned)xx < (unsigned)W) ? ip[yy * W + xx] : 0.f;
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KPER; ++kk) {
                    float wv[RRP];
#pragma unroll
                    for (int t = 0; t < RRP; t += 4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(&wl[kk][c][t]);      // wave-uniform address: broadcast
                        wv[t] = w4.x; wv[t + 1] = w4.y; wv[t + 2] = w4.z; wv[t + 3] = w4.w;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < R; ++s2)
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(wv[r * R + s2], iv[r][s2 + q], acc[kk][q]);
                }
            }
        }
        LC_STAMP(4);
        for (int j = 0; j < J; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(&tyl[wave][j][0]);
            const float tysum = fabsf(t4.x) + fabsf(t4.y) + fabsf(t4.z) + fabsf(t4.w);
            if (__builtin_amdgcn_readfirstlane(__float_as_uint(tysum)) == 0u) continue;      // the row misses the object (uniform)
            float txs[R][4];
            float txsum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < R; ++s2)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = px(xq, q) * stride + s2 - pad;
                    txs[s2][q] = ((unsigned)xx < (unsigned)Wm) ? txp[j][xx] : 0.f;
                    txsum += fabsf(txs[s2][q]);
                }
            if (__ballot(txsum != 0.f) == 0ull) continue;
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk) {
                const float4 c4 = csl[wave][j][kk];
                const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int s2 = 0; s2 < R; ++s2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(cs[s2], txs[s2][q], acc[kk][q]);
            }
        }
        LC_STAMP(5);
#pragma unroll
        for (int kk = 0; kk < KPER; ++kk) {
            if (k0 + kk >= K) continue;
            float* orow = out + (((size_t)n * K + k0 + kk) * OH + oy) * OW;
            if (vec) {
                *reinterpret_cast<float4*>(orow + xq * 4) = make_float4(acc[kk][0], acc[kk][1], acc[kk][2], acc[kk][3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (px(xq, q) < OW) orow[px(xq, q)] = acc[kk][q];
            }
        }
    }
    LC_STAMP(3);
    }
}

// Third form (3 x 3, pad 1, stride 1, no pooling, map rows of <= 256 pixels in whole quads, K % KPER == 0, <= 3 image channels):
// what conv_dim_in / down_flow[0] and the dense J = 0 calls launch.  The kernel above, measured with in-kernel clocks
// (tools/ubench/lc_stamps.hip) on the c3 shape: a channel block takes a wave ~24 k cycles (43 k with image channels) -- 3 k building the
// taps, 5 k + 19 k accumulator set-up and image channels, 10 k walking the 22 object slots, 5-8 k issuing 16 stores -- against the 13 k the
// 16 KB of stores need at the 5.9 TB/s this store pattern reaches alone (tools/ubench/store_pattern.hip).  Every global load in
// the block loop (the U entries under a condition, the filter block, the image window -- the SAME 54 values for every block) was
// followed by s_waitcnt vmcnt(0), which on this ISA also waits for the previous block's stores: compute and stores never overlapped.
// Here nothing in the loop waits for memory behind a store:
//   * the image rows the workgroup's four output rows see are staged ONCE, scaled and zero-padded, in LDS (imgl);
//   * U[n, :, k0 : k0 + KPER] and the filter block of the NEXT channel block are copied global -> LDS by LDS-DMA pieces
//     (buffer_load ... lds, no registers) issued after the barrier that frees the buffers; the copy is waited for (vmcnt(0)) just
//     BEFORE the current block's stores are issued, when only loads are outstanding, and published by the barrier at the next top;
//   * each wave keeps a list of the objects its row touches (built once): the taps are contracted for those only and the object
//     loop walks the list (no per-slot LDS round trip + branch); an object's six profile values per lane are one 16-byte read + 2.
// A row that touches more than ACT objects contracts the extra ones on the fly (correct, slower; c3 rows touch 2-5).
typedef __attribute__((address_space(3))) void* lc_lds_vptr;
// Sums of 16 per-lane values over the 64 lanes of a wave with 17 exchanges instead of 96: four halving steps (each lane keeps the half of
// its values that its side of the exchange owns and adds the partner's), then two full steps.  Lane l returns the total of value (l >> 2) & 15.
__device__ __forceinline__ float lc_wave_sum16(const float (&v)[16]) {
    const int lane = threadIdx.x & 63;
    float a[8], b[4], c[2];
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (h5 ? v[i + 8] : v[i]) + __shfl_xor(h5 ? v[i] : v[i + 8], 32, 64);      // value i + 8 bit5
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (h4 ? a[i + 4] : a[i]) + __shfl_xor(h4 ? a[i] : a[i + 4], 16, 64);      // i + 4 bit4 + 8 bit5
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = (h3 ? b[i + 2] : b[i]) + __shfl_xor(h3 ? b[i] : b[i + 2], 8, 64);       // i + 2 bit3 + ...
    float r = (h2 ? c[1] : c[0]) + __shfl_xor(h2 ? c[0] : c[1], 4, 64);                                        // bit2 + 2 bit3 + 4 bit4 + 8 bit5
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    return r;
}
template <int KPER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void layout_conv_fwd3_kernel(const float* __restrict__ U, const float* __restrict__ boxes, const int* __restrict__ keep,
                             const float* __restrict__ residual, float* __restrict__ out, int J, int K, int H, int W,
                             const float* __restrict__ img, const float* __restrict__ wimg, const float* __restrict__ iscale, int IC,
                             float* __restrict__ stat_partial, int Ng) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the stub; it rejects the LDS-DMA builtin)
    constexpr int R = 3, RR = 9, RRP = 12, TJ = LC_TJ, TW = LC_TW, ACT = 10, UB = KPER * RR, IR = R + 3, IWP = 272;
    static_assert(KPER == 16, "the row statistics reduce 16 channels across the wave");
    __shared__ float2 rowst[4][KPER];             // (mean, M2) of the wave's row for each channel of the block (stat_partial != null)
    constexpr int UPIECES = (TJ * UB * 4 + 1023) / 1024, WPIECES = (KPER * 3 * RR * 4 + 1023) / 1024;
    __shared__ float txp[TJ][TW];                 // column profiles; index x + 4 (x = -4 .. 259 for W <= 256)
    __shared__ float tyl[4][TJ][4];
    __shared__ int actl[4][TJ + 1];               // [wave][0] = number of objects the wave's row touches, then their slots
    __shared__ float4 csl[4][ACT][KPER];
    __shared__ float wl[KPER][3][RRP];
    __shared__ __attribute__((aligned(16))) float ul[UPIECES * 256];      // DMA target [j][KPER][RR] (+ the last piece's tail)
    __shared__ __attribute__((aligned(16))) float wraw[WPIECES * 256];    // DMA target [KPER][IC][RR]
    __shared__ __attribute__((aligned(16))) float imgl[3][IR][IWP];       // scaled image rows oy0 - 1 .. oy0 + 4; index x + 4
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int oy0 = blockIdx.x * 4, oy = oy0 + wave;
    const int n = blockIdx.y;
    const bool has_img = img != nullptr;
    // ---- once per workgroup: profiles, the rows' taps and object lists, the image rows
    for (int idx = threadIdx.x; idx < J * TW; idx += 256) {
        const int j = idx / TW, xi = idx - j * TW, xx = xi - 4;
        const float* b = boxes + ((size_t)n * J + j) * 4;
        txp[j][xi] = ((unsigned)xx < (unsigned)W) ? lc_profile(xx, W, 0, b[0], b[2]) : 0.f;
    }
    {
        bool touched = false;
        if (lane < J) {
            const float* b = boxes + ((size_t)n * J + lane) * 4;
            const float bx = b[0], by = b[1], bw = b[2], bh = b[3];
            const bool act = ((bx != 0.f) || (by != 0.f) || (bw != 0.f) || (bh != 0.f)) && keep[(size_t)n * J + lane] != 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int yy = oy + r - 1;
                const float v = (r < R && oy < H && act && (unsigned)yy < (unsigned)H) ? lc_profile(yy, H, 0, by, bh) : 0.f;
                tyl[wave][lane][r] = v;
                touched = touched || v != 0.f;
            }
        }
        const unsigned long long m = __ballot(touched);
        if (lane == 0) actl[wave][0] = __popcll(m);
        if (touched) actl[wave][1 + __popcll(m & ((1ull << lane) - 1ull))] = lane;
    }
    if (has_img) {
        const float sc = iscale ? iscale[n] : 1.f;
        for (int idx = threadIdx.x; idx < 3 * IR * IWP; idx += 256) {
            const int xi = idx % IWP, ri = (idx / IWP) % IR, c = idx / (IWP * IR);
            const int xx = xi - 4, yy = oy0 - 1 + ri;
            (&imgl[0][0][0])[idx] = (c < IC && (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H)
                                        ? img[(((size_t)n * IC + c) * H + yy) * W + xx] * sc : 0.f;
        }
    }
    // ---- LDS-DMA of a channel block's U entries and filter block
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U ? U + (size_t)n * J * K * RR : out), 0,
                                                                        U ? (unsigned)((size_t)J * K * RR * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_img ? wimg : out), 0,
                                                                        has_img ? (unsigned)((size_t)K * IC * RR * 4) : 0u, 0x00020000);
    unsigned uoff[(UPIECES + 3) / 4];
#pragma unroll
    for (int i = 0; i < (UPIECES + 3) / 4; ++i) {
        const int e = ((wave + 4 * i) * 64 + lane) * 4;            // first float of this lane's 16 bytes in ul
        const int j = e / UB, rem = e - j * UB;
        uoff[i] = (wave + 4 * i < UPIECES && j < J) ? (unsigned)((j * K * RR + rem) * 4) : 0xFFFFFFF0u;      // out of range: zeros
    }
    const unsigned woff = (wave < WPIECES && (wave * 64 + lane) * 4 < KPER * IC * RR) ? (unsigned)((wave * 64 + lane) * 16) : 0xFFFFFFF0u;
    auto dma = [&](int kb) {
#pragma unroll
        for (int i = 0; i < (UPIECES + 3) / 4; ++i)
            if (wave + 4 * i < UPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (lc_lds_vptr)(&ul[(wave + 4 * i) * 256]), 16, uoff[i], (unsigned)(kb * RR * 4), 0, 0);
        if (wave < WPIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lc_lds_vptr)(&wraw[wave * 256]), 16, woff, (unsigned)(kb * IC * RR * 4), 0, 0);
    };
    dma(blockIdx.z * KPER);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int xq = lane;
    const bool px_live = oy < H && xq * 4 < W;
    // batch-norm statistics of the output (stat_partial != null): one Chan partial (count, mean, M2) per (sample, channel, 4-row band),
    // laid out as bn_stats_finalize_kernel reads them -- the consumer's statistics pass over the [N, K, H, W] result is not needed
    auto band_partial = [&](int kprev) {
        if (threadIdx.x < KPER) {
            float cnt = 0.f, mean = 0.f, m2 = 0.f;
            for (int r = 0; r < 4 && oy0 + r < H; ++r) {
                const float2 p = rowst[r][threadIdx.x];
                const float nb = (float)W, nt = cnt + nb, d = p.x - mean;
                mean += d * (nb / nt);
                m2 += p.y + d * d * (cnt * nb / nt);
                cnt = nt;
            }
            const int nbands = gridDim.x, g = n / Ng, nl = n - g * Ng;
            float* o = stat_partial + (((size_t)g * K + kprev + threadIdx.x) * ((size_t)Ng * nbands) + (size_t)nl * nbands + blockIdx.x) * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        }
    };
    for (int k0 = blockIdx.z * KPER; k0 < K; k0 += gridDim.z * KPER) {
        __syncthreads();          // the block's U / filter copies have landed (every wave waited for its pieces); csl / wl of the previous block are free
        if (stat_partial && k0 != blockIdx.z * KPER) band_partial(k0 - gridDim.z * KPER);      // (rowst is rewritten behind the next barrier)
        const int na = actl[wave][0];
        for (int idx = lane; idx < min(na, ACT) * KPER; idx += 64) {
            const int a = idx / KPER, kk = idx - a * KPER;
            const int j = actl[wave][1 + a];
            const float4 t4 = *reinterpret_cast<const float4*>(&tyl[wave][j][0]);
            const float* u = &ul[j * UB + kk * RR];
            float cs[3];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) cs[s2] = fmaf(t4.z, u[6 + s2], fmaf(t4.y, u[3 + s2], t4.x * u[s2]));
            csl[wave][a][kk] = make_float4(cs[0], cs[1], cs[2], 0.f);
        }
        if (has_img) {
            for (int idx = threadIdx.x; idx < KPER * 3 * RRP; idx += 256) {
                const int t = idx % RRP, c = (idx / RRP) % 3, kk = idx / (RRP * 3);
                (&wl[0][0][0])[idx] = (t < RR && c < IC) ? wraw[(kk * IC + c) * RR + t] : 0.f;
            }
        }
        __syncthreads();          // ul / wraw are read out: the next block's copies may start (unless a row has extra objects, which read ul below)
        const bool extra = (actl[0][0] > ACT) || (actl[1][0] > ACT) || (actl[2][0] > ACT) || (actl[3][0] > ACT);      // uniform over the workgroup
        const int kn = k0 + gridDim.z * KPER;
        if (!extra && kn < K) dma(kn);
        float acc[KPER][4];
        if (px_live) {
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[kk][q] = residual ? residual[(((size_t)n * K + k0 + kk) * H + oy) * W + xq * 4 + q] : 0.f;
            if (has_img) {
                for (int c = 0; c < IC; ++c) {
                    float iv[R][6];                 // columns 4 xq - 1 .. 4 xq + 4
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float* row = &imgl[c][wave + r][xq * 4 + 4];
                        const float4 m4 = *reinterpret_cast<const float4*>(row);
                        iv[r][0] = row[-1]; iv[r][1] = m4.x; iv[r][2] = m4.y; iv[r][3] = m4.z; iv[r][4] = m4.w; iv[r][5] = row[4];
                    }
#pragma unroll
                    for (int kk = 0; kk < KPER; ++kk) {
                        float wv[RRP];
#pragma unroll
                        for (int t = 0; t < RRP; t += 4) {
                            const float4 w4 = *reinterpret_cast<const float4*>(&wl[kk][c][t]);      // wave-uniform address: broadcast
                            wv[t] = w4.x; wv[t + 1] = w4.y; wv[t + 2] = w4.z; wv[t + 3] = w4.w;
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int s2 = 0; s2 < R; ++s2)
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(wv[r * R + s2], iv[r][s2 + q], acc[kk][q]);
                    }
                }
            }
            for (int a = 0; a < na; ++a) {
                const int j = __builtin_amdgcn_readfirstlane(actl[wave][1 + a]);
                const float* row = &txp[j][xq * 4 + 4];
                const float4 m4 = *reinterpret_cast<const float4*>(row);
                const float tx[6] = {row[-1], m4.x, m4.y, m4.z, m4.w, row[4]};
                const float txsum = fabsf(tx[0]) + fabsf(tx[1]) + fabsf(tx[2]) + fabsf(tx[3]) + fabsf(tx[4]) + fabsf(tx[5]);
                if (__ballot(txsum != 0.f) == 0ull) continue;
                if (a < ACT) {
#pragma unroll
                    for (int kk = 0; kk < KPER; ++kk) {
                        const float4 c4 = csl[wave][a][kk];
                        const float cs[3] = {c4.x, c4.y, c4.z};
#pragma unroll
                        for (int s2 = 0; s2 < R; ++s2)
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(cs[s2], tx[s2 + q], acc[kk][q]);
                    }
                } else {            // more objects on this row than the tap table holds: contract on the fly
                    const float4 t4 = *reinterpret_cast<const float4*>(&tyl[wave][j][0]);
                    for (int kk = 0; kk < KPER; ++kk) {
                        const float* u = &ul[j * UB + kk * RR];
#pragma unroll
                        for (int s2 = 0; s2 < R; ++s2) {
                            const float cs = fmaf(t4.z, u[6 + s2], fmaf(t4.y, u[3 + s2], t4.x * u[s2]));
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[kk][q] = fmaf(cs, tx[s2 + q], acc[kk][q]);
                        }
                    }
                }
            }
        }
        if (extra) {              // (rare) the object loop read ul: copy only now
            __syncthreads();
            if (kn < K) dma(kn);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the copies (and nothing younger: this block's stores follow)
        if (px_live) {
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk)
                *reinterpret_cast<float4*>(out + (((size_t)n * K + k0 + kk) * H + oy) * W + xq * 4) =
                    make_float4(acc[kk][0], acc[kk][1], acc[kk][2], acc[kk][3]);
        }
        if (stat_partial && oy < H) {        // (wave-uniform) behind the stores: two passes over the row held in registers
            float v[KPER];
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk) v[kk] = px_live ? (acc[kk][0] + acc[kk][1]) + (acc[kk][2] + acc[kk][3]) : 0.f;
            const float tot = lc_wave_sum16(v);               // lane l: channel (l >> 2) & 15
            const float invw = 1.f / (float)W;
            float mu[KPER];
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk)                                                                // (wave-uniform: scalar registers)
                mu[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 4 * kk)) * invw;
#pragma unroll
            for (int kk = 0; kk < KPER; ++kk) {
                const float d0 = acc[kk][0] - mu[kk], d1 = acc[kk][1] - mu[kk], d2 = acc[kk][2] - mu[kk], d3 = acc[kk][3] - mu[kk];
                v[kk] = px_live ? fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3) : 0.f;
            }
            const float m2 = lc_wave_sum16(v);
            if ((lane & 3) == 0) rowst[wave][lane >> 2] = make_float2(tot * invw, m2);      // (lane l holds channel l >> 2 of both sums)
        }
    }
    if (stat_partial) {
        __syncthreads();
        const int nblk = (K / KPER - 1 - (int)blockIdx.z) / (int)gridDim.z;      // index of this workgroup's last channel block
        band_partial((int)blockIdx.z * KPER + nblk * (int)gridDim.z * KPER);
    }
#endif
}

// dU[n,j,k,r,s] = sum_{oy,ox} dout[n,k,oy,ox] * ty_j(oy*st+r-pad) * tx_j(ox*st+s-pad): grid (J, NB, ceil(K/32)),
// wave w reduces channels k = 32*blockIdx.z + w + 4*i; only the rows / 64-column blocks the object touches are read.
template <int R>
__global__ __launch_bounds__(256) void layout_conv_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                              const int* __restrict__ keep, float* __restrict__ dU, int J, int K,
                                                              int H, int W, int Hm, int Wm, int pool, int OH, int OW, int stride,
                                                              int pad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x, n = blockIdx.y;
    const float* b = boxes + ((size_t)n * J + j) * 4;
    const float x0 = b[0], y0 = b[1], ww = b[2], hh = b[3];
    const bool legal = (x0 != 0.f) || (y0 != 0.f) || (ww != 0.f) || (hh != 0.f);
    const bool active = legal && keep[(size_t)n * J + j] != 0;
    // Support of the object in output coordinates, found once per wave: rows / 64-column blocks where some tap is nonzero.
    int oy_lo = OH, oy_hi = -1;
    unsigned long long xblk_mask = 0;   // bit b: column block [32b, 32b+32) has a nonzero tap (OW <= 2048)
    if (active) {
        for (int base = 0; base < OH; base += 64) {
            const int oy = base + lane;
            float tyr[R];
            lc_taps_y<R>(tyr, oy, stride, pad, Hm, H, pool, y0, hh, oy < OH);
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) t += fabsf(tyr[r]);
            const unsigned long long b = __ballot(t != 0.f);
            if (b) {
                oy_lo = min(oy_lo, base + (int)__builtin_ctzll(b));
                oy_hi = max(oy_hi, base + 63 - (int)__builtin_clzll(b));
            }
        }
        for (int xb = 0; xb < OW && xb < 2048; xb += 64) {
            const int ox = xb + lane;
            float t = 0.f;
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const int xx = ox * stride + s - pad;
                t += (ox < OW && (unsigned)xx < (unsigned)Wm) ? fabsf(lc_profile(xx, W, pool, x0, ww)) : 0.f;
            }
            const unsigned long long bal = __ballot(t != 0.f);
            if (bal & 0xFFFFFFFFull) xblk_mask |= 1ull << (xb >> 5);
            if (bal >> 32) xblk_mask |= 2ull << (xb >> 5);
        }
    }
    // The wave's 8 channels (k = 32 z + wave + 4 i) go through the support together, CH at a time: the row taps are evaluated once
    // per row instead of once per (row, channel), and a row issues CH independent loads.
    constexpr int CH = (R == 3) ? 8 : 4;
    const size_t plane = (size_t)OH * OW;
    for (int c0 = 0; c0 < 8; c0 += CH) {
        const int kfirst = blockIdx.z * 32 + wave + 4 * c0;
        if (kfirst >= K) break;
        float acc[CH][R][R];
#pragma unroll
        for (int ci = 0; ci < CH; ++ci)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int s = 0; s < R; ++s) acc[ci][r][s] = 0.f;
        if (active && oy_hi >= oy_lo) {
            // a wave pass covers 32 columns x 2 rows (128-byte row segments): the support is rounded to 32 columns, not 64 -- an object
            // 77 pixels wide touches 3-4 such blocks instead of 2-3 blocks of 64 (the pass read x2.28 of the support's bytes)
            for (int xb = 0; xb < OW; xb += 32) {
                if (xb < 2048 && !((xblk_mask >> (xb >> 5)) & 1ull)) continue;   // wave-uniform
                const int ox = xb + (lane & 31);
                float txs[R];
                float txsum = 0.f;
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const int xx = ox * stride + s - pad;
                    txs[s] = (ox < OW && (unsigned)xx < (unsigned)Wm) ? lc_profile(xx, W, pool, x0, ww) : 0.f;
                    txsum += fabsf(txs[s]);
                }
                const bool col_live = txsum != 0.f && ox < OW;
                const float* dp = dout + ((size_t)n * K + kfirst) * plane + ox;
                for (int oyb = oy_lo; oyb <= oy_hi; oyb += 2) {
                    const int oy = oyb + (lane >> 5);
                    const bool live = col_live && oy <= oy_hi;
                    float g[CH];
#pragma unroll
                    for (int ci = 0; ci < CH; ++ci)
                        g[ci] = (live && kfirst + 4 * ci < K) ? dp[(size_t)(4 * ci) * plane + (size_t)oy * OW] : 0.f;
                    float tyr[R];
                    lc_taps_y<R>(tyr, oy, stride, pad, Hm, H, pool, y0, hh, live);
                    float w9[R][R];
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int s = 0; s < R; ++s) w9[r][s] = tyr[r] * txs[s];
#pragma unroll
                    for (int ci = 0; ci < CH; ++ci)
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int s = 0; s < R; ++s) acc[ci][r][s] += g[ci] * w9[r][s];
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < CH; ++ci) {
            const int k = kfirst + 4 * ci;
            if (k >= K) break;
            float* o = dU + (((size_t)n * J + j) * K + k) * (R * R);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int s = 0; s < R; ++s) {
                    const float v = wave_sum(acc[ci][r][s]);
                    if (lane == 0) o[r * R + s] = v;
                }
        }
    }
}

// ---------------------------------------------------------------- flow warp
struct WarpTap { int x0, y0, x1, y1; float fx, fy, mx, my; };
__device__ __forceinline__ WarpTap warp_tap(float fxv, float fyv, int x, int y, int H, int W) {
    // get_grid: linspace(-1,1); flow normalised by (W-1)/2, (H-1)/2; unnormalise with align_corners=False; border clip
    auto lin11 = [](int i, int steps) -> float {
        if (steps == 1) return -1.f;
        const float step = 2.0f / (float)(steps - 1);
        return (i < steps / 2) ? -1.0f + step * (float)i : 1.0f - step * (float)(steps - 1 - i);
    };
    const float gx = lin11(x, W) + fxv / (((float)W - 1.0f) / 2.0f);
    const float gy = lin11(y, H) + fyv / (((float)H - 1.0f) / 2.0f);
    float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    float iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    WarpTap t;
    t.mx = 1.f; t.my = 1.f;
    if (ix <= 0.f) { ix = 0.f; t.mx = 0.f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); t.mx = 0.f; }
    if (iy <= 0.f) { iy = 0.f; t.my = 0.f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); t.my = 0.f; }
    const float flx = floorf(ix), fly = floorf(iy);
    t.x0 = (int)flx; t.y0 = (int)fly; t.x1 = t.x0 + 1; t.y1 = t.y0 + 1;
    t.fx = ix - flx; t.fy = iy - fly;
    return t;
}

__global__ __launch_bounds__(256) void flow_warp_fwd_kernel(const float* __restrict__ img, const float* __restrict__ flow,
                                                            float* __restrict__ out, int N, int C, int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * H * W;
    if (idx >= total) return;
    const int HW = H * W;
    const int n = (int)(idx / HW);
    const int p = (int)(idx - (long long)n * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + (size_t)n * 2 * HW;
    const WarpTap t = warp_tap(fl[p], fl[HW + p], x, y, H, W);
    const bool x1ok = t.x1 < W, y1ok = t.y1 < H;
    const float wnw = (1.f - t.fx) * (1.f - t.fy), wne = t.fx * (1.f - t.fy), wsw = (1.f - t.fx) * t.fy, wse = t.fx * t.fy;
    for (int c = 0; c < C; ++c) {
        const float* ip = img + ((size_t)n * C + c) * HW;
        float v = ip[t.y0 * W + t.x0] * wnw;
        if (x1ok) v += ip[t.y0 * W + t.x1] * wne;
        if (y1ok) v += ip[t.y1 * W + t.x0] * wsw;
        if (x1ok && y1ok) v += ip[t.y1 * W + t.x1] * wse;
        out[((size_t)n * C + c) * HW + p] = v;
    }
}

// dflow always; dimg (pre-zeroed, atomics) when non-null
__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ img,
                                                            const float* __restrict__ flow, float* __restrict__ dflow,
                                                            float* __restrict__ dimg, int N, int C, int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * H * W;
    if (idx >= total) return;
    const int HW = H * W;
    const int n = (int)(idx / HW);
    const int p = (int)(idx - (long long)n * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + (size_t)n * 2 * HW;
    const WarpTap t = warp_tap(fl[p], fl[HW + p], x, y, H, W);
    const bool x1ok = t.x1 < W, y1ok = t.y1 < H;
    const float wnw = (1.f - t.fx) * (1.f - t.fy), wne = t.fx * (1.f - t.fy), wsw = (1.f - t.fx) * t.fy, wse = t.fx * t.fy;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t pl = ((size_t)n * C + c) * HW;
        const float g = dout[pl + p];
        const float* ip = img + pl;
        const float vnw = ip[t.y0 * W + t.x0];
        const float vne = x1ok ? ip[t.y0 * W + t.x1] : 0.f;
        const float vsw = y1ok ? ip[t.y1 * W + t.x0] : 0.f;
        const float vse = (x1ok && y1ok) ? ip[t.y1 * W + t.x1] : 0.f;
        gix += g * ((vne - vnw) * (1.f - t.fy) + (vse - vsw) * t.fy);
        giy += g * ((vsw - vnw) * (1.f - t.fx) + (vse - vne) * t.fx);
        if (dimg) {
            float* dp = dimg + pl;
            atomicAdd(dp + t.y0 * W + t.x0, g * wnw);
            if (x1ok) atomicAdd(dp + t.y0 * W + t.x1, g * wne);
            if (y1ok) atomicAdd(dp + t.y1 * W + t.x0, g * wsw);
            if (x1ok && y1ok) atomicAdd(dp + t.y1 * W + t.x1, g * wse);
        }
    }
    // d(ix)/d(flow_x) = (W/2) * clip' / ((W-1)/2)
    float* df = dflow + (size_t)n * 2 * HW;
    df[p] = gix * t.mx * ((float)W * 0.5f) / (((float)W - 1.0f) / 2.0f);
    df[HW + p] = giy * t.my * ((float)H * 0.5f) / (((float)H - 1.0f) / 2.0f);
}

// ---------------------------------------------------------------- avg_pool2d(3, stride 2, pad 1, count_include_pad=False)
__global__ __launch_bounds__(256) void avgpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total,
                                                             int H, int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long long r = idx / OW;
    const int oy = (int)(r % OH);
    const long long nc = r / OH;
    const float* xp = x + (size_t)nc * H * W;
    float s = 0.f;
    int cnt = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = 2 * oy - 1 + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = 2 * ox - 1 + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            s += xp[iy * W + ix];
            ++cnt;
        }
    }
    y[idx] = s / (float)cnt;
}

__device__ __forceinline__ int pool_cnt(int o, int L) {  // valid taps of window o along a length-L axis
    int c = 0;
    for (int d = 0; d < 3; ++d) { const int i = 2 * o - 1 + d; c += ((unsigned)i < (unsigned)L) ? 1 : 0; }
    return c;
}
__global__ __launch_bounds__(256) void avgpool3s2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long total,
                                                             int H, int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % W);
    const long long r = idx / W;
    const int iy = (int)(r % H);
    const long long nc = r / H;
    const float* dp = dy + (size_t)nc * OH * OW;
    float s = 0.f;
    // windows oy with 2*oy-1 <= iy <= 2*oy+1
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
        if (oy >= OH) continue;
        const int cy = pool_cnt(oy, H);
        for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
            if (ox >= OW) continue;
            s += dp[oy * OW + ox] / (float)(cy * pool_cnt(ox, W));
        }
    }
    dx[idx] = s;
}

// ---------------------------------------------------------------- nearest resize (gather) and its adjoint
__device__ __forceinline__ int nearest_src(int o, int in, int out) {
    if (in == out) return o;
    if (out == 2 * in) return o >> 1;
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)o * scale);
    return s < in - 1 ? s : in - 1;
}
__global__ __launch_bounds__(256) void nearest_resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total,
                                                                 int H, int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long long r = idx / OW;
    const int oy = (int)(r % OH);
    const long long nc = r / OH;
    y[idx] = x[(size_t)nc * H * W + (size_t)nearest_src(oy, H, OH) * W + nearest_src(ox, W, OW)];
}
// Integer down-sampling factors (the decoder's segmentation-map pyramid, normalization.py:102: 128^2 -> 64^2 -> ... of a [120, 512, ., .]
// tensor) and 2x up-sampling (fx < 0), output rows in whole quads: one thread per 4 outputs -- 32-bit index arithmetic, 16-byte loads / stores (the per-element form
// above spent its time in 64-bit div / mod chains: 3.45 ms per c3 step for 7 GB of traffic).  Words are copied as they are (split words too).
__global__ __launch_bounds__(256) void nearest_down_quads_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned total_q,
                                                                 int H, int W, int OH, int OW, int fy, int fx) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total_q) return;
    const unsigned qw = (unsigned)OW >> 2, qp = (unsigned)OH * qw;
    const unsigned nc = idx / qp, rem = idx - nc * qp;
    const unsigned oy = rem / qw, q = rem - oy * qw;
    float4 o;
    if (fx < 0) {             // 2x up-sampling (the flow network's decoder, flows_generator.py: nn.Upsample(scale_factor=2)): two inputs per quad
        const float2 a = *reinterpret_cast<const float2*>(x + (size_t)nc * H * W + (size_t)(oy >> 1) * W + (size_t)q * 2);
        reinterpret_cast<float4*>(y)[idx] = make_float4(a.x, a.x, a.y, a.y);
        return;
    }
    const float* src = x + (size_t)nc * H * W + (size_t)(oy * fy) * W + (size_t)q * 4 * fx;
    if (fx == 1) {
        o = *reinterpret_cast<const float4*>(src);
    } else if (fx == 2) {
        const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
        o = make_float4(a.x, a.z, b.x, b.z);
    } else {
        o = make_float4(src[0], src[fx], src[2 * fx], src[3 * fx]);
    }
    reinterpret_cast<float4*>(y)[idx] = o;
}
// adjoint: dx[iy,ix] = sum of dy over all outputs that read (iy,ix).  One thread per INPUT element; the candidate outputs
// are found by scanning the (small) preimage range, so no atomics and no memset are needed.
__global__ __launch_bounds__(256) void nearest_resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long total,
                                                                 int H, int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % W);
    const long long r = idx / W;
    const int iy = (int)(r % H);
    const long long nc = r / H;
    const float* dp = dy + (size_t)nc * OH * OW;
    if (H % OH == 0 && W % OW == 0 && H >= OH && W >= OW) {
        // integer down-sampling factor f: output o reads input o*f, so only inputs on the f-grid receive a gradient
        const int fy = H / OH, fx = W / OW;
        dx[idx] = (iy % fy == 0 && ix % fx == 0) ? dp[(iy / fy) * OW + (ix / fx)] : 0.f;
        return;
    }
    if (OH == 2 * H && OW == 2 * W) {  // nearest x2 up-sampling: each input is read by a 2x2 output block
        const float* q = dp + (size_t)(2 * iy) * OW + 2 * ix;
        dx[idx] = (q[0] + q[1]) + (q[OW] + q[OW + 1]);
        return;
    }
    // outputs o with nearest_src(o) == i lie in [i*out/in - 1, (i+1)*out/in + 1]
    const int oy_lo = max(0, (int)((long long)iy * OH / H) - 1), oy_hi = min(OH - 1, (int)((long long)(iy + 1) * OH / H) + 1);
    const int ox_lo = max(0, (int)((long long)ix * OW / W) - 1), ox_hi = min(OW - 1, (int)((long long)(ix + 1) * OW / W) + 1);
    float s = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        if (nearest_src(oy, H, OH) != iy) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox)
            if (nearest_src(ox, W, OW) == ix) s += dp[oy * OW + ox];
    }
    dx[idx] = s;
}

// adjoint of the 2x up-sampling in pairs: dx[iy, 2 j .. 2 j + 1] from one 16-byte piece of each of the two output rows (same sums, same order
// as the per-element form above)
__global__ __launch_bounds__(256) void nearest_up2_bwd_pairs_kernel(const float* __restrict__ dy, float* __restrict__ dx, unsigned total_p,
                                                                    int H, int W) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total_p) return;
    const unsigned pw = (unsigned)W >> 1, pp = (unsigned)H * pw;
    const unsigned nc = idx / pp, rem = idx - nc * pp;
    const unsigned iy = rem / pw, j = rem - iy * pw;
    const float* q = dy + (size_t)nc * 4 * H * W + (size_t)(2 * iy) * (2 * W) + (size_t)j * 4;
    const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 2 * W);
    reinterpret_cast<float2*>(dx)[idx] = make_float2((a.x + a.y) + (b.x + b.y), (a.z + a.w) + (b.z + b.w));
}

// Adjoint of a nearest-neighbour pyramid: one full-resolution map x[NC,H,W] was down-sampled (integer factors) to up to 8
// levels, each used by some SPADE layers (normalization.py:102 -- F.interpolate(segmap, size, 'nearest') in every SPADE).
// dx = g_full + sum_l scatter(g_l): level l with factors (fy, fx) sends g_l[y, x] to dx[y*fy, x*fx].  One pass: every dx
// element is written once (4 consecutive x per thread), instead of one full-size scatter + one full-size add per level.
struct PyramidLevels {
    const float* g[8];
    int h[8], w[8], fy[8], fx[8];
    int n;
};
__global__ __launch_bounds__(256) void nearest_pyramid_bwd_kernel(const float* __restrict__ g_full, PyramidLevels lv,
                                                                  float* __restrict__ dx, long long total4, int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of dx
    if (idx >= total4) return;
    const int W4 = W >> 2;
    const int x0 = (int)(idx % W4) * 4;
    const long long r = idx / W4;
    const int y = (int)(r % H);
    const long long nc = r / H;
    float4 v = g_full ? reinterpret_cast<const float4*>(g_full)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* ve = reinterpret_cast<float*>(&v);
    for (int l = 0; l < lv.n; ++l) {
        const int fy = lv.fy[l], fx = lv.fx[l];
        if (y % fy) continue;
        const float* gp = lv.g[l] + ((size_t)nc * lv.h[l] + y / fy) * lv.w[l];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + e;
            if (x % fx == 0) ve[e] += gp[x / fx];
        }
    }
    reinterpret_cast<float4*>(dx)[idx] = v;
}

// ================================================================ C ABI
extern "C" int ag2v_layout_render_fwd(const float* vecs, const float* boxes, const int* keep, float* out, int BT, int O, int D,
                                      int H, int W, void* stream) {
    if (!vecs || !boxes || !keep || !out || BT <= 0 || O <= 0 || O > MAXO || D <= 0 || H <= 0 || W <= 0 || (W & 3)) return AG2V_EINVAL;
    if (BT > 65535) return AG2V_EINVAL;
    int dsplit = 1;
    const int rows = ceil_div(H, 4);
    while ((long long)rows * BT * dsplit < 2048 && dsplit < 16 && D / (dsplit * 2) >= 8) dsplit *= 2;
    const int dper = ceil_div(D, dsplit);
    hipLaunchKernelGGL(layout_render_fwd_kernel, dim3(rows, BT, dsplit), dim3(256), 0, (hipStream_t)stream, vecs, boxes, keep, out,
                       O, D, H, W, dper);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_layout_render_bwd(const float* dout, const float* boxes, const int* keep, float* dvecs, int BT, int O, int D,
                                      int H, int W, void* stream) {
    if (!dout || !boxes || !keep || !dvecs || BT <= 0 || O <= 0 || O > MAXO || D <= 0 || H <= 0 || W <= 0) return AG2V_EINVAL;
    if (BT > 65535) return AG2V_EINVAL;
    hipLaunchKernelGGL(layout_render_bwd_kernel, dim3(O, BT, ceil_div(D, 32)), dim3(256), 0, (hipStream_t)stream, dout, boxes,
                       keep, dvecs, O, D, H, W);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// The launch conditions of layout_conv_fwd3_kernel that do not depend on addresses (U / wimg must also be 16-byte aligned)
static bool lc_fwd3_shape_ok(int J, int K, int H, int W, int R, int stride, int pad, int pool, int IC, bool has_img) {
    static const bool no_tables = false;       // A/B aids
    static const bool no_fwd3 = false;
    return J <= LC_TJ && W <= LC_TW && !no_tables && !no_fwd3 && R == 3 && pad == 1 && stride == 1 && pool == 0 && W <= 256 && (W & 3) == 0 &&
           K % 16 == 0 && (!has_img || IC <= 3) && (size_t)J * K * 9 * 4 < 0xFFFFFFF0ull;
}
// 1 when ag2v_layout_conv_img_fwd_stats can write batch-norm partials for this call (16-byte aligned U / wimg provided)
extern "C" int ag2v_layout_conv_stats_supported(int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, int IC, int Ng) {
    static const bool off = getenv("AG2V_NO_LC_STATS") != nullptr;
    return (!off && NB > 0 && Ng > 0 && NB % Ng == 0 && NB <= 65535 && lc_fwd3_shape_ok(J, K, H, W, R, stride, pad, pool, IC, IC > 0)) ? 1 : 0;
}
// floats of the partial buffer: [NB / Ng, K, Ng * ceil(H / 4), 3]
extern "C" size_t ag2v_layout_conv_stats_floats(int NB, int K, int H) { return (size_t)NB * K * ceil_div(H, 4) * 3; }
static int layout_conv_img_fwd_impl(const float* U, const float* boxes, const int* keep, const float* residual, float* out,
                                    int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, const float* img,
                                    const float* wimg, const float* iscale, int IC, float* stat_partial, int Ng, void* stream) {
    // J == 0 (with img): no layout term at all -- a same-size convolution of <= 8 dense channels to K outputs through the kernel's
    // image path (U / boxes / keep are not read): the data gradients of conv_img / conv_flow, whose gathered tensor has 3 / 2 channels
    if (!out || NB <= 0 || J < 0 || (J == 0 && !img) || (J > 0 && (!U || !boxes || !keep)) || K <= 0 || H <= 0 || W <= 0 || stride <= 0 ||
        NB > 65535) return AG2V_EINVAL;
    if ((R != 3 && R != 4) || pool < 0 || pool > 1) return AG2V_EINVAL;
    const int Hm = pool ? (H - 1) / 2 + 1 : H, Wm = pool ? (W - 1) / 2 + 1 : W;   // size of the map the convolution sees
    const int OH = (Hm + 2 * pad - R) / stride + 1, OW = (Wm + 2 * pad - R) / stride + 1;
    if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
    if (img) {      // the fused image channels: same-size stride-1 convolution, 16-byte rows (the kernel's 4-pixel layout)
        if (!wimg || IC <= 0 || IC > 8 || stride != 1 || pool != 0 || (OW & 3) || OH != H || OW != W) return AG2V_EINVAL;
    }
    constexpr int KPER = 16;
    dim3 grid(ceil_div(OH, 4), NB, ceil_div(K, KPER));
    static const bool no_tables = false;       // A/B aid
    const bool tables = J <= LC_TJ && Wm <= LC_TW && !no_tables;
#define LC_LAUNCH(KERNEL, RR) hipLaunchKernelGGL((KERNEL<RR, KPER>), grid, dim3(256), 0, (hipStream_t)stream, U, boxes, keep, residual, \
                                                 out, J, K, H, W, Hm, Wm, pool, OH, OW, stride, pad, img, wimg, iscale, IC)
    if (tables) {       // channel blocks per workgroup: all of them once rows x images alone fill the chip several times over
        const int nkb = ceil_div(K, KPER), wg = ceil_div(OH, 4) * NB;
        grid.z = wg >= 2048 ? 1 : std::min(nkb, ceil_div(2048, wg));
    }
    const bool fwd3 = tables && lc_fwd3_shape_ok(J, K, H, W, R, stride, pad, pool, IC, img != nullptr) &&
                      (((uintptr_t)U | (uintptr_t)wimg) & 15) == 0;
    if (stat_partial && (!fwd3 || Ng <= 0 || NB % Ng != 0)) return AG2V_EINVAL;      // statistics come out of that kernel only
    if (fwd3) {
        hipLaunchKernelGGL((layout_conv_fwd3_kernel<KPER>), grid, dim3(256), 0, (hipStream_t)stream, U, boxes, keep, residual, out, J, K, H, W,
                           img, wimg, iscale, IC, stat_partial, Ng > 0 ? Ng : 1);
    } else
    if (tables) { if (R == 3) LC_LAUNCH(layout_conv_fwd_kernel, 3); else LC_LAUNCH(layout_conv_fwd_kernel, 4); }
    else        { if (R == 3) LC_LAUNCH(layout_conv_fwd_generic_kernel, 3); else LC_LAUNCH(layout_conv_fwd_generic_kernel, 4); }
#undef LC_LAUNCH
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_layout_conv_img_fwd(const float* U, const float* boxes, const int* keep, const float* residual, float* out,
                                        int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, const float* img,
                                        const float* wimg, const float* iscale, int IC, void* stream) {
    return layout_conv_img_fwd_impl(U, boxes, keep, residual, out, NB, J, K, H, W, R, stride, pad, pool, img, wimg, iscale, IC, nullptr, 1, stream);
}
// ... and the batch-norm statistics of `out` as Chan partials (count, mean, M2) per (sample, channel, band of 4 rows), in the layout
// ag2v_bn_finalize_partials reads: stat_partial[NB / Ng][K][Ng * ceil(H / 4)][3] (ag2v_layout_conv_stats_floats), for a batch norm whose
// statistics are taken per group of Ng consecutive samples (generator.py:30 conv_dim_in -> SynchronizedBatchNorm2d).  Only where
// ag2v_layout_conv_stats_supported() says so; AG2V_EINVAL otherwise.
extern "C" int ag2v_layout_conv_img_fwd_stats(const float* U, const float* boxes, const int* keep, const float* residual, float* out,
                                              int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, const float* img,
                                              const float* wimg, const float* iscale, int IC, float* stat_partial, int Ng, void* stream) {
    if (!stat_partial) return AG2V_EINVAL;
    return layout_conv_img_fwd_impl(U, boxes, keep, residual, out, NB, J, K, H, W, R, stride, pad, pool, img, wimg, iscale, IC, stat_partial, Ng,
                                    stream);
}
extern "C" int ag2v_layout_conv_fwd(const float* U, const float* boxes, const int* keep, const float* residual, float* out,
                                    int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, void* stream) {
    return ag2v_layout_conv_img_fwd(U, boxes, keep, residual, out, NB, J, K, H, W, R, stride, pad, pool, nullptr, nullptr, nullptr, 0,
                                    stream);
}

extern "C" int ag2v_layout_conv_bwd(const float* dout, const float* boxes, const int* keep, float* dU, int NB, int J, int K, int H,
                                    int W, int R, int stride, int pad, int pool, void* stream) {
    if (!dout || !boxes || !keep || !dU || NB <= 0 || J <= 0 || K <= 0 || H <= 0 || W <= 0 || stride <= 0 || NB > 65535) return AG2V_EINVAL;
    if ((R != 3 && R != 4) || pool < 0 || pool > 1) return AG2V_EINVAL;
    const int Hm = pool ? (H - 1) / 2 + 1 : H, Wm = pool ? (W - 1) / 2 + 1 : W;
    const int OH = (Hm + 2 * pad - R) / stride + 1, OW = (Wm + 2 * pad - R) / stride + 1;
    if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
    dim3 grid(J, NB, ceil_div(K, 32));
    if (R == 3)
        hipLaunchKernelGGL(layout_conv_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, dout, boxes, keep, dU, J, K, H, W, Hm,
                           Wm, pool, OH, OW, stride, pad);
    else
        hipLaunchKernelGGL(layout_conv_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dout, boxes, keep, dU, J, K, H, W, Hm,
                           Wm, pool, OH, OW, stride, pad);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_flow_warp_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream) {
    if (!img || !flow || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return AG2V_EINVAL;
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(flow_warp_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, img, flow, out, N, C, H, W);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_flow_warp_bwd(const float* dout, const float* img, const float* flow, float* dflow, float* dimg, int N, int C,
                                  int H, int W, void* stream) {
    if (!dout || !img || !flow || !dflow || N <= 0 || C <= 0 || H <= 0 || W <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dimg && hipMemsetAsync(dimg, 0, (size_t)N * C * H * W * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, dout, img, flow, dflow, dimg, N, C, H, W);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// ---------------------------------------------------------------- max_pool2d(2, stride 2) (VGG19 features, architecture.py:93-123)
// floor mode: OH = H / 2, OW = W / 2; a trailing odd row / column belongs to no window
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int H,
                                                           int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long long r = idx / OW;
    const int oy = (int)(r % OH);
    const long long nc = r / OH;
    const float* xp = x + (size_t)nc * H * W + (size_t)(2 * oy) * W + 2 * ox;
    y[idx] = fmaxf(fmaxf(xp[0], xp[1]), fmaxf(xp[W], xp[W + 1]));
}
// one thread per INPUT element: it receives its window's gradient iff it is the window's first maximum in row-major order
// (ATen's max_pool2d backward routes the gradient to the index found with a strict '>' scan)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dx, long long total, int H, int W, int OH, int OW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % W);
    const long long r = idx / W;
    const int iy = (int)(r % H);
    const long long nc = r / H;
    const int oy = iy >> 1, ox = ix >> 1;
    float g = 0.f;
    if (oy < OH && ox < OW) {
        const float* xp = x + (size_t)nc * H * W + (size_t)(2 * oy) * W + 2 * ox;
        const float v[4] = {xp[0], xp[1], xp[W], xp[W + 1]};
        int arg = 0;
        float m = v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (v[k] > m) { m = v[k]; arg = k; }
        if (arg == ((iy & 1) << 1 | (ix & 1))) g = dy[(size_t)nc * OH * OW + (size_t)oy * OW + ox];
    }
    dx[idx] = g;
}
extern "C" int ag2v_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, void* stream) {
    if (!x || !y || NC <= 0 || H < 2 || W < 2) return AG2V_EINVAL;
    const int OH = H / 2, OW = W / 2;
    const long long total = (long long)NC * OH * OW;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_maxpool2_bwd(const float* dy, const float* x, float* dx, int NC, int H, int W, void* stream) {
    if (!dy || !x || !dx || NC <= 0 || H < 2 || W < 2) return AG2V_EINVAL;
    const int OH = H / 2, OW = W / 2;
    const long long total = (long long)NC * H * W;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_avgpool3s2_fwd(const float* x, float* y, int NC, int H, int W, void* stream) {
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0) return AG2V_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long total = (long long)NC * OH * OW;
    hipLaunchKernelGGL(avgpool3s2_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_avgpool3s2_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream) {
    if (!dy || !dx || NC <= 0 || H <= 0 || W <= 0) return AG2V_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long total = (long long)NC * H * W;
    hipLaunchKernelGGL(avgpool3s2_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_nearest_resize_fwd(const float* x, float* y, int NC, int H, int W, int OH, int OW, void* stream) {
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return AG2V_EINVAL;
    const long long total = (long long)NC * OH * OW;
    const bool down = OH <= H && OW <= W && H % OH == 0 && W % OW == 0, up2 = OH == 2 * H && OW == 2 * W;
    if ((down || up2) && (OW & 3) == 0 && total / 4 < (1LL << 31) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
        true) {
        const unsigned tq = (unsigned)(total / 4);
        hipLaunchKernelGGL(nearest_down_quads_kernel, dim3((tq + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, x, y, tq, H, W, OH, OW,
                           up2 ? 0 : H / OH, up2 ? -2 : W / OW);
        AG2V_CHECK_LAUNCH();
        return AG2V_OK;
    }
    hipLaunchKernelGGL(nearest_resize_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_nearest_resize_bwd(const float* dy, float* dx, int NC, int H, int W, int OH, int OW, void* stream) {
    if (!dy || !dx || NC <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return AG2V_EINVAL;
    const long long total = (long long)NC * H * W;
    if (OH == 2 * H && OW == 2 * W && (W & 1) == 0 && total / 2 < (1LL << 31) && (((uintptr_t)dy & 15) | ((uintptr_t)dx & 7)) == 0 &&
        true) {
        const unsigned tp = (unsigned)(total / 2);
        hipLaunchKernelGGL(nearest_up2_bwd_pairs_kernel, dim3((tp + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, dy, dx, tp, H, W);
        AG2V_CHECK_LAUNCH();
        return AG2V_OK;
    }
    hipLaunchKernelGGL(nearest_resize_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, total, H, W, OH, OW);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// dx[NC,H,W] = g_full (nullable) + sum over levels of the nearest-neighbour scatter of g_levels[l] ([NC, hs[l], ws[l]]; H % hs[l] == 0,
// W % ws[l] == 0; null entries are skipped).  g_levels / hs / ws are HOST arrays of n_levels (<= 8) entries.
extern "C" int ag2v_nearest_pyramid_bwd(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                                        float* dx, int NC, int H, int W, void* stream) {
    if (!dx || NC <= 0 || H <= 0 || W <= 0 || (W & 3) || n_levels < 0 || n_levels > 8 || (n_levels && (!g_levels || !hs || !ws)))
        return AG2V_EINVAL;
    PyramidLevels lv{};
    for (int l = 0; l < n_levels; ++l) {
        if (!g_levels[l]) continue;
        if (hs[l] <= 0 || ws[l] <= 0 || H % hs[l] || W % ws[l]) return AG2V_EINVAL;
        lv.g[lv.n] = g_levels[l]; lv.h[lv.n] = hs[l]; lv.w[lv.n] = ws[l]; lv.fy[lv.n] = H / hs[l]; lv.fx[lv.n] = W / ws[l];
        ++lv.n;
    }
    const long long total4 = (long long)NC * H * (W >> 2);
    hipLaunchKernelGGL(nearest_pyramid_bwd_kernel, dim3(ceil_div(total4, 256)), dim3(256), 0, (hipStream_t)stream, g_full, lv, dx,
                       total4, H, W);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
