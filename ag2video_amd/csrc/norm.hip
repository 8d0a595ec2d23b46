// Normalisation kernels of the AG2Video hot path (HBM-bound):
//   * batch statistics (per channel over N,H,W) with a robust two-pass-in-registers + Chan combine
//       reference: F.batch_norm fallback of _SynchronizedBatchNorm (sync_batchnorm/batchnorm.py:63-68)
//   * fused  BN(affine) -> LeakyReLU            (flow net / conv_dim_in: normalization.py:39-46, generator.py:26-33)
//   * fused  SPADE: BN(no affine) * (1+gamma) + beta [-> LeakyReLU]   (normalization.py:96-110, architecture.py:53-68)
//   * InstanceNorm2d(affine=False) -> LeakyReLU (discriminator.py:374 via normalization.py:43-44)
//   and their backward passes, plus the per-channel sum used for conv bias gradients.
// All tensors NCHW fp32 contiguous.  One workgroup owns one <=4096-element chunk of one (n,c) plane, so the
// per-channel parameters are scalars and every access is a coalesced float4 (or dword when H*W % 4 != 0).
#include "common.h"
#include "conv_common.h"   // pack_split()

#define CHUNK 4096  // elements per workgroup (16 per thread)

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }

// load up to 16 elements/thread of plane chunk [start, start+cnt) into v[], returns #valid via mask bits
template <bool VEC>
__device__ __forceinline__ void load16(const float* __restrict__ p, int cnt, float (&v)[16], unsigned& mask) {
    mask = 0;
    if (VEC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (i * 256 + threadIdx.x) * 4;
            if (e < cnt) {
                const float4 t = *reinterpret_cast<const float4*>(p + e);
                v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
                mask |= 0xFu << (4 * i);
            } else {
                v[4 * i] = v[4 * i + 1] = v[4 * i + 2] = v[4 * i + 3] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = i * 256 + threadIdx.x;
            if (e < cnt) { v[i] = p[e]; mask |= 1u << i; } else v[i] = 0.f;
        }
    }
}
template <bool VEC>
__device__ __forceinline__ void store16(float* __restrict__ p, int cnt, const float (&v)[16]) {
    if (VEC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (i * 256 + threadIdx.x) * 4;
            if (e < cnt) *reinterpret_cast<float4*>(p + e) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = i * 256 + threadIdx.x;
            if (e < cnt) p[e] = v[i];
        }
    }
}

// ---------------------------------------------------------------- batch statistics
// partial[(c*nparts + part)*3 + {0,1,2}] = {count, mean, M2}
// Sample groups (frame-batched generator: statistics per frame group of Ng samples, as the T separate F.batch_norm calls of the
// reference's frame loop compute them): "channel" gc = (n / Ng) * C + c has nparts = Ng * nchunk parts.  Ng == N: plain batch norm.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               int Ng, int C, int HW, int nchunk) {
    __shared__ float sh[4];
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int n = nc / C, c = nc - n * C;
    const int start = chunk * CHUNK;
    const int cnt = min(CHUNK, HW - start);
    float v[16];
    unsigned mask;
    load16<VEC>(x + (size_t)nc * HW + start, cnt, v, mask);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    s = block_sum_256(s, sh);
    const float mean = s / (float)cnt;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float d = v[i] - mean;
        m2 += ((mask >> i) & 1u) ? d * d : 0.f;
    }
    m2 = block_sum_256(m2, sh);
    if (threadIdx.x == 0) {
        const int nparts = Ng * nchunk;
        const int g = n / Ng, nl = n - g * Ng;
        float* o = partial + ((size_t)(g * C + c) * nparts + (size_t)nl * nchunk + chunk) * 3;
        o[0] = (float)cnt; o[1] = mean; o[2] = m2;
    }
}

// One thread per channel: Chan's parallel combine in double per group; writes mean / invstd (biased var, +eps) of every group and
// applies the G momentum updates of the running statistics IN GROUP ORDER, exactly as G consecutive
// F.batch_norm(training=True, momentum) calls do (unbiased var in the running buffer).
// grid: ceil(C / 16) blocks of 256 threads = 16 channels x 16 group lanes (lane gl handles groups gl, gl + 16, ...).
#define FIN_MAXG 64
// rep: every element stands for `rep` equal elements (statistics of a nearest-upsampled tensor taken on its source: counts and M2 x rep);
// running_mean2 / running_var2 (nullable): a second pair of running buffers receiving the same updates (two norm layers on one input).
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partial, int G, int C, int nparts, float eps,
                                                                float momentum, float* __restrict__ mean_out,
                                                                float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, int rep = 1,
                                                                float* __restrict__ running_mean2 = nullptr,
                                                                float* __restrict__ running_var2 = nullptr) {
    __shared__ float s_mean[16][FIN_MAXG], s_unb[16][FIN_MAXG];
    const int cl = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + cl;
    if (c < C) {
        for (int g = gl; g < G; g += 16) {
            const float* p = partial + (size_t)(g * C + c) * nparts * 3;
            double n = 0.0, mean = 0.0, m2 = 0.0;
            for (int i = 0; i < nparts; ++i) {
                const double nb = (double)p[3 * i] * rep, mb = p[3 * i + 1], m2b = (double)p[3 * i + 2] * rep;
                const double nt = n + nb;
                const double d = mb - mean;
                mean += d * (nb / nt);
                m2 += m2b + d * d * (n * nb / nt);
                n = nt;
            }
            const double var = m2 / n;
            mean_out[g * C + c] = (float)mean;
            invstd_out[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            if (g < FIN_MAXG) {
                s_mean[cl][g] = (float)mean;
                s_unb[cl][g] = (float)((n > 1.0) ? m2 / (n - 1.0) : var);
            }
        }
    }
    __syncthreads();
    if (c < C && gl == 0 && running_mean) {
        double rm = (double)running_mean[c], rv = (double)running_var[c];
        for (int g = 0; g < G; ++g) {       // each update rounds to fp32 as the separate calls would
            rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)s_mean[cl][g]);
            rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)s_unb[cl][g]);
        }
        running_mean[c] = (float)rm;
        running_var[c] = (float)rv;
    }
    if (c < C && gl == 0 && running_mean2) {
        double rm = (double)running_mean2[c], rv = (double)running_var2[c];
        for (int g = 0; g < G; ++g) {
            rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)s_mean[cl][g]);
            rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)s_unb[cl][g]);
        }
        running_mean2[c] = (float)rm;
        running_var2[c] = (float)rv;
    }
}

// ---------------------------------------------------------------- fused normalise (+affine | +SPADE) (+LeakyReLU)
// MODE 0: y = lrelu((x-mean)*invstd*w[c] + b[c]);  MODE 1: y = lrelu((x-mean)*invstd*(1+gamma) + beta)
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int C,
                                                         int HW, int nchunk, float slope, long long gbx, int Ng, int out_split,
                                                         float* __restrict__ yhalf = nullptr, int W = 0) {
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int c = nc % C;
    const int gc = ((nc / C) / Ng) * C + c;          // statistics of this sample's group
    const int start = chunk * CHUNK;
    const int cnt = min(CHUNK, HW - start);
    const size_t off = (size_t)nc * HW + start;
    const float mu = mean[gc], is = invstd[gc];
    float v[16], g[16], be[16];
    unsigned mask;
    load16<VEC>(x + off, cnt, v, mask);
    if (MODE == 1) {
        const size_t goff = off + (size_t)(nc / C) * gbx;   // gbx = C*HW when gamma | beta are the halves of one [N,2C,HW] tensor
        load16<VEC>(gamma + goff, cnt, g, mask);
        load16<VEC>(beta + goff, cnt, be, mask);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lrelu_f((v[i] - mu) * is * (1.f + g[i]) + be[i], slope);
    } else {
        const float sc = is * (w ? w[c] : 1.f), sh = b ? b[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lrelu_f((v[i] - mu) * sc + sh, slope);
    }
    if (out_split) {        // the output is read by bf16x3 convolutions only: store it as split words (conv_common.h pack_split)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = pack_split(v[i]);
    }
    store16<VEC>(y + off, cnt, v);
    if (VEC && yhalf) {
        // the 2x nearest-neighbour down-sampling of y (the first level of the decoder's segmentation-map pyramid: y[2i][2j]) from the
        // registers: W % 8 == 0, so a thread's four consecutive elements share a row and start at a multiple of 4
        const int Wh = W >> 1;
        float* yh = yhalf + (size_t)nc * (HW >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (i * 256 + threadIdx.x) * 4;
            if (e >= cnt) continue;
            const int L = start + e;
            const int row = L / W, x0 = L - row * W;
            if (row & 1) continue;
            *reinterpret_cast<float2*>(yh + (size_t)(row >> 1) * Wh + (x0 >> 1)) = make_float2(v[4 * i], v[4 * i + 2]);
        }
    }
}

// Gradient of a tensor that was also read through nearest-neighbour down-samplings (the segmentation-map pyramid of the decoder):
// dy = g_full + sum_l scatter(g_l).  The affine batch-norm backward can form it while loading (PYR) instead of reading a
// materialised sum that a separate pass wrote: level l with factors (fy, fx) contributes g_l[y / fy, x / fx] at on-grid (y, x).
struct NormPyramid {
    const float* g[8];
    int h[8], w[8], fy[8], fx[8];
    int n;
    int W;      // row length of the full-resolution map
    int pow2;   // every factor is a power of two
};
// adds the level terms to the 16 elements load16<true> fetched for (image plane nc, elements start ...) of a [., H, W] tensor
__device__ __forceinline__ void pyramid_add16(float (&g)[16], const NormPyramid& lv, size_t nc, int start, int cnt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4;
        if (e >= cnt) continue;
        const int L = start + e;
        const int y = L / lv.W, x0 = L - y * lv.W;       // (W % 4 == 0: the four elements share the row)
        int l0 = 0;
        if (lv.pow2) {
            // the first two levels (after ops' fold of the small ones: all of them) are FETCHED before either is added: every level handled
            // by the loop below is a load whose use sits behind the next level's branch, i.e. one more dependent memory round trip
            float v[2][4];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                v[l][0] = v[l][1] = v[l][2] = v[l][3] = 0.f;
                if (l >= lv.n) continue;
                const int fy = lv.fy[l], fx = lv.fx[l];
                if (y & (fy - 1)) continue;
                const int sy = 31 - __clz(fy), sx = 31 - __clz(fx);
                const float* gp = lv.g[l] + (nc * lv.h[l] + (y >> sy)) * lv.w[l] + (x0 >> sx);
                if (fx == 2) { v[l][0] = gp[0]; v[l][2] = gp[1]; }
                else if (fx == 1) { v[l][0] = gp[0]; v[l][1] = gp[1]; v[l][2] = gp[2]; v[l][3] = gp[3]; }
                else if ((x0 & (fx - 1)) == 0) v[l][0] = gp[0];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) g[4 * i + q] += v[0][q] + v[1][q];
            l0 = 2;
        }
        for (int l = l0; l < lv.n; ++l) {
            const int fy = lv.fy[l], fx = lv.fx[l];
            if (lv.pow2) {          // power-of-two factors (the decoder's pyramid): masks and shifts, no integer divisions
                if (y & (fy - 1)) continue;
                const int sy = 31 - __clz(fy), sx = 31 - __clz(fx);
                const float* gp = lv.g[l] + (nc * lv.h[l] + (y >> sy)) * lv.w[l] + (x0 >> sx);
                if (fx == 2) { g[4 * i] += gp[0]; g[4 * i + 2] += gp[1]; }
                else if (fx == 1) { g[4 * i] += gp[0]; g[4 * i + 1] += gp[1]; g[4 * i + 2] += gp[2]; g[4 * i + 3] += gp[3]; }
                else if ((x0 & (fx - 1)) == 0) g[4 * i] += gp[0];
                continue;
            }
            if (y % fy) continue;
            const float* gp = lv.g[l] + (nc * lv.h[l] + y / fy) * lv.w[l];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = x0 + q;
                if (x % fx == 0) g[4 * i + q] += gp[x / fx];
            }
        }
    }
}
// CHUNKs a workgroup of a pure-read reduce pass walks before its one block reduction: as many as keep >= 16 workgroups per CU in the
// launch (AG2V_REDUCE_CPB overrides, for A/B runs)
static int reduce_chunks_per_block(long long planes, int nchunk) {
    constexpr int forced = 0;
    if (forced > 0) return forced < nchunk ? forced : nchunk;
    int cpb = 1;
    while (cpb < 8 && cpb * 2 <= nchunk && planes * ceil_div(nchunk, cpb * 2) >= 16LL * 256) cpb *= 2;
    return cpb;
}
// backward pass 1: per (c, part) partial sums  s1 = sum dxhat, s2 = sum dxhat * xhat
//   MODE 0: dxhat = g (weight applied in pass 2);  MODE 1: dxhat = g * (1 + gamma);   g = dy * lrelu'(y)
template <int MODE, bool VEC, bool PYR = false>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ partial, int Ng, int C, int HW, int nchunk,
                                                              float slope, long long gbx, const float* __restrict__ aw,
                                                              const float* __restrict__ ab, int cpb, NormPyramid pyr = NormPyramid{}) {
    // nchunk = workgroups per plane; a workgroup walks cpb consecutive CHUNKs and reduces once at the end (measured: 18.6 -> 17.7 ms for
    // the c3 pyramid node's pair of passes, nothing on the SPADE reduce passes: the block reductions were not what these passes wait for)
    __shared__ float sh[16];
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int n = nc / C, c = nc - n * C;
    const int grp = n / Ng, nl = n - grp * Ng, gc = grp * C + c;
    const float mu = mean[gc], is = invstd[gc];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int it = 0; it < cpb; ++it) {
        const int start = (chunk * cpb + it) * CHUNK;
        if (start >= HW) break;
        const int cnt = min(CHUNK, HW - start);
        const size_t off = (size_t)nc * HW + start;
        float g[16], xv[16], t[16];
        unsigned mask;
        if (PYR && dy == nullptr) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = 0.f;
            mask = 0;
        } else
            load16<VEC>(dy + off, cnt, g, mask);
        if (PYR) pyramid_add16(g, pyr, (size_t)nc, start, cnt);
        load16<VEC>(x + off, cnt, xv, mask);
        if (slope != 1.f) {
            if (MODE == 0 && y == nullptr) {
                // affine batch norm: the activation's sign is the sign of (x - mu) * (invstd w) + b, the expression norm_apply_kernel<0>
                // evaluated -- no read of the forward output (2 of this backward's 7 tensor passes)
                const float sc = is * (aw ? aw[c] : 1.f), shf = ab ? ab[c] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = ((xv[i] - mu) * sc + shf) > 0.f ? g[i] : g[i] * slope;
            } else {
                load16<VEC>(y + off, cnt, t, mask);
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = t[i] > 0.f ? g[i] : g[i] * slope;
            }
        }
        // (s3, s4) = (sum g, sum g * xhat) BEFORE the (1 + gamma) factor: the channel sums of dbeta and dgamma, i.e. the bias gradient of the
        // convolution that produced gamma | beta -- it would otherwise cost a pass over the two [N, C, HW] gradient tensors (channel_sum)
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s3 += g[i];
                s4 += g[i] * ((xv[i] - mu) * is);
            }
            load16<VEC>(gamma + off + (size_t)n * gbx, cnt, t, mask);
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] *= (1.f + t[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s1 += g[i];  // invalid lanes hold g == 0
            s2 += g[i] * ((xv[i] - mu) * is);
        }
    }
    if (MODE == 1) {
        float v4[4] = {s1, s2, s3, s4};
        block_sums_256<4>(v4, sh);
        s1 = v4[0]; s2 = v4[1]; s3 = v4[2]; s4 = v4[3];
    } else {
        float v2[2] = {s1, s2};
        block_sums_256<2>(v2, sh);
        s1 = v2[0]; s2 = v2[1];
    }
    if (threadIdx.x == 0) {
        const int nparts = Ng * nchunk;
        float* o = partial + ((size_t)gc * nparts + (size_t)nl * nchunk + chunk) * 4;
        o[0] = s1; o[1] = s2; o[2] = s3; o[3] = s4;
    }
}

// sums[c*4+{0..3}] = sum over parts (double accumulate)
__global__ void pair_sum_finalize_kernel(const float* __restrict__ partial, int C, int nparts, float* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = partial + (size_t)c * nparts * 4;
    double a = 0.0, b = 0.0, d = 0.0, e = 0.0;
    for (int i = 0; i < nparts; ++i) { a += p[4 * i]; b += p[4 * i + 1]; d += p[4 * i + 2]; e += p[4 * i + 3]; }
    sums[4 * c] = (float)a;
    sums[4 * c + 1] = (float)b;
    sums[4 * c + 2] = (float)d;
    sums[4 * c + 3] = (float)e;
}

// backward pass 2.
//   MODE 0: dx = w*invstd*(g - s1/n - xhat*s2/n)                    (dw = s2, db = s1 are read from sums by the host)
//   MODE 1: dgamma = g*xhat, dbeta = g, dx = invstd*(g(1+gamma) - s1/n - xhat*s2/n)
template <int MODE, bool VEC, bool PYR = false>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ w, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ sums,
                                                             float* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int C, int HW, int nchunk,
                                                             float inv_count, float slope, long long gbx, int Ng, int dgb_split,
                                                             const float* __restrict__ ab, long long ggx = -1,
                                                             NormPyramid pyr = NormPyramid{}) {
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int c = nc % C;
    const int gc = ((nc / C) / Ng) * C + c;
    const int start = chunk * CHUNK;
    const int cnt = min(CHUNK, HW - start);
    const size_t off = (size_t)nc * HW + start;
    const float mu = mean[gc], is = invstd[gc];
    const float m1 = sums[4 * gc] * inv_count, m2 = sums[4 * gc + 1] * inv_count;
    float g[16], xv[16], t[16];
    unsigned mask;
    if (PYR && dy == nullptr) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = 0.f;
        mask = 0;
    } else
    load16<VEC>(dy + off, cnt, g, mask);
    if (PYR) pyramid_add16(g, pyr, (size_t)nc, start, cnt);
    load16<VEC>(x + off, cnt, xv, mask);
    if (slope != 1.f) {
        if (MODE == 0 && y == nullptr) {       // see norm_bwd_reduce_kernel
            const float sc = is * (w ? w[c] : 1.f), shf = ab ? ab[c] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = ((xv[i] - mu) * sc + shf) > 0.f ? g[i] : g[i] * slope;
        } else {
            load16<VEC>(y + off, cnt, t, mask);
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = t[i] > 0.f ? g[i] : g[i] * slope;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = (xv[i] - mu) * is;  // xhat
    if (MODE == 1) {
        const size_t goff = off + (size_t)(nc / C) * gbx;
        // (ggx >= 0: gamma kept as a compact [N, C, HW] copy -- the lean tape -- while dgamma | dbeta stay halves of [N, 2C, HW])
        load16<VEC>(gamma + (ggx >= 0 ? off + (size_t)(nc / C) * ggx : goff), cnt, t, mask);
        float dg[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dg[i] = g[i] * xv[i];
        if (dgb_split & 1) {    // dgamma | dbeta go to a bf16x3 data gradient + weight gradient only: stored as split words
            float ps[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { dg[i] = pack_split(dg[i]); ps[i] = pack_split(g[i]); }
            store16<VEC>(dgamma + goff, cnt, dg);
            store16<VEC>(dbeta + goff, cnt, ps);
        } else {
            store16<VEC>(dgamma + goff, cnt, dg);
            store16<VEC>(dbeta + goff, cnt, g);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = is * (g[i] * (1.f + t[i]) - m1 - xv[i] * m2);
    } else {
        const float sc = is * (w ? w[c] : 1.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = sc * (g[i] - m1 - xv[i] * m2);
    }
    if (dgb_split & 4) {        // dx goes to the data / weight gradient (+ bias sum) of the convolution that produced x, and nothing else
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = pack_split(g[i]);
    }
    store16<VEC>(dx + off, cnt, g);
}

// ---------------------------------------------------------------- SPADE pair on one (optionally 2x nearest-upsampled) input
// A SPADEResnetBlock with a learned shortcut normalises the SAME tensor twice (norm_0 and norm_s, architecture.py:52-53,60-61): same
// batch statistics, different gamma | beta.  Its input is the previous block's output after nn.Upsample(scale_factor=2)
// (spade_generator.py:62-76).  This family evaluates both modulations in one pass over x and reads x at its ORIGINAL resolution
// (UP = 2: element (y, x) of the plane is xs[y / 2][x / 2]), so the upsampled tensor is never written or read, the statistics pass
// runs on a quarter of the elements (replicating every element four times changes neither mean nor biased variance; the element
// count of the unbiased variance is scaled by `rep`), and the backward writes ONE input gradient -- the sum of both branches, already
// reduced over each 2 x 2 replica block -- instead of two full-resolution tensors, their sum and the upsample's adjoint.
template <int UP, bool VEC>
__device__ __forceinline__ void load16_x(const float* __restrict__ xs_plane, int start, int cnt, int W, float (&v)[16]) {
    if (UP == 1) {
        unsigned mask;
        load16<VEC>(xs_plane + start, cnt, v, mask);
        return;
    }
    const int Ws = W >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (i * 256 + threadIdx.x) * 4;
        if (e < cnt) {
            const int o = start + e;
            const int oy = o / W, ox = o - oy * W;          // W % 4 == 0: the four elements share a row
            const float2 t = *reinterpret_cast<const float2*>(xs_plane + (size_t)(oy >> 1) * Ws + (ox >> 1));
            v[4 * i] = t.x; v[4 * i + 1] = t.x; v[4 * i + 2] = t.y; v[4 * i + 3] = t.y;
        } else {
            v[4 * i] = v[4 * i + 1] = v[4 * i + 2] = v[4 * i + 3] = 0.f;
        }
    }
}

// y0 = lrelu(xhat (1 + g0) + b0, slope0), y1 = lrelu(xhat (1 + g1) + b1, slope1); gb_k = [N, 2C, HW] (gamma half, then beta half)
template <int UP>
__global__ __launch_bounds__(256) void spade_pair_apply_kernel(const float* __restrict__ xs, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gb0,
                                                               const float* __restrict__ gb1, float* __restrict__ y0,
                                                               float* __restrict__ y1, int C, int HW, int W, int nchunk, float slope0,
                                                               float slope1, int Ng, int split0, int split1) {
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int n = nc / C, c = nc - n * C;
    const int gc = (n / Ng) * C + c;
    const int start = chunk * CHUNK;
    const int cnt = min(CHUNK, HW - start);
    const size_t off = (size_t)nc * HW + start;
    const size_t goff = ((size_t)n * 2 * C + c) * HW + start;
    const float mu = mean[gc], is = invstd[gc];
    float v[16], g[16], be[16], o[16];
    unsigned mask;
    load16_x<UP, true>(xs + (size_t)nc * (HW / (UP * UP)), start, cnt, W, v);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (v[i] - mu) * is;
    load16<true>(gb0 + goff, cnt, g, mask);
    load16<true>(gb0 + goff + (size_t)C * HW, cnt, be, mask);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = lrelu_f(v[i] * (1.f + g[i]) + be[i], slope0);
    if (split0) {       // read by a bf16x3 convolution only: stored as split words (ag2v_conv_opts::in_split)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = pack_split(o[i]);
    }
    store16<true>(y0 + off, cnt, o);
    load16<true>(gb1 + goff, cnt, g, mask);
    load16<true>(gb1 + goff + (size_t)C * HW, cnt, be, mask);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = lrelu_f(v[i] * (1.f + g[i]) + be[i], slope1);
    if (split1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = pack_split(o[i]);
    }
    store16<true>(y1 + off, cnt, o);
}

// per (group, channel, part): 8 sums = for k in {0, 1}: (sum dxhat_k, sum dxhat_k xhat, sum g_k, sum g_k xhat)   (cf. norm_bwd_reduce_kernel)
template <int UP>
__global__ __launch_bounds__(256) void spade_pair_bwd_reduce_kernel(const float* __restrict__ dy0, const float* __restrict__ y0,
                                                                    const float* __restrict__ dy1, const float* __restrict__ y1,
                                                                    const float* __restrict__ xs, const float* __restrict__ gb0,
                                                                    const float* __restrict__ gb1, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, float* __restrict__ partial, int Ng,
                                                                    int C, int HW, int W, int nchunk, float slope0, float slope1,
                                                                    int gcompact, int cpb) {
    // nchunk = workgroups per plane, each walking cpb consecutive CHUNKs (see norm_bwd_reduce_kernel)
    __shared__ float sh[32];
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int n = nc / C, c = nc - n * C;
    const int grp = n / Ng, nl = n - grp * Ng, gc = grp * C + c;
    const float mu = mean[gc], is = invstd[gc];
    float out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = 0.f;
    for (int it = 0; it < cpb; ++it) {
        const int start = (chunk * cpb + it) * CHUNK;
        if (start >= HW) break;
        const int cnt = min(CHUNK, HW - start);
        const size_t off = (size_t)nc * HW + start;
        const size_t goff = ((size_t)n * 2 * C + c) * HW + start;
        float xh[16], g[16], t[16];
        unsigned mask;
        load16_x<UP, true>(xs + (size_t)nc * (HW / (UP * UP)), start, cnt, W, xh);
#pragma unroll
        for (int i = 0; i < 16; ++i) xh[i] = (xh[i] - mu) * is;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* dy = k ? dy1 : dy0;
            const float* y = k ? y1 : y0;
            const float* gb = k ? gb1 : gb0;
            const float slope = k ? slope1 : slope0;
            load16<true>(dy + off, cnt, g, mask);
            if (slope != 1.f) {
                load16<true>(y + off, cnt, t, mask);
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = t[i] > 0.f ? g[i] : g[i] * slope;
            }
            load16<true>(gb + (gcompact ? off : goff), cnt, t, mask);     // gcompact: gamma as a compact [N, C, HW] copy (lean tape)
            float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool live = (mask >> i) & 1u;             // (xhat of a lane beyond the plane is -mu * is, not 0)
                const float gi = live ? g[i] : 0.f, xi = live ? xh[i] : 0.f;
                s3 += gi;
                s4 += gi * xi;
                const float d = gi * (1.f + t[i]);
                s1 += d;
                s2 += d * xi;
            }
            out[4 * k] += s1; out[4 * k + 1] += s2; out[4 * k + 2] += s3; out[4 * k + 3] += s4;
        }
    }
    block_sums_256<8>(out, sh);
    if (threadIdx.x == 0) {
        const int nparts = Ng * nchunk;
        float* o = partial + ((size_t)gc * nparts + (size_t)nl * nchunk + chunk) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = out[k];
    }
}
// sums[c * width + k] = sum over parts (double accumulate)
__global__ void multi_sum_finalize_kernel(const float* __restrict__ partial, int C, int nparts, int width, float* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * width) return;
    const int c = i / width, k = i - c * width;
    const float* p = partial + (size_t)c * nparts * width + k;
    double a = 0.0;
    for (int j = 0; j < nparts; ++j) a += p[(size_t)j * width];
    sums[i] = (float)a;
}
// dgb_k = (g_k xhat | g_k);  dx = invstd * sum_k (g_k (1 + gamma_k) - m1_k - xhat m2_k), for UP = 2 summed over each 2 x 2 replica block
// and written at the input's own resolution.  UP = 2 needs whole row pairs per chunk: CHUNK % (2 W) == 0 (checked by the launcher).
template <int UP>
__global__ __launch_bounds__(256) void spade_pair_bwd_apply_kernel(const float* __restrict__ dy0, const float* __restrict__ y0,
                                                                   const float* __restrict__ dy1, const float* __restrict__ y1,
                                                                   const float* __restrict__ xs, const float* __restrict__ gb0,
                                                                   const float* __restrict__ gb1, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ sums,
                                                                   float* __restrict__ dxs, float* __restrict__ dgb0,
                                                                   float* __restrict__ dgb1, int C, int HW, int W, int nchunk,
                                                                   float inv_count, float slope0, float slope1, int Ng, int dgb_split,
                                                                   int gcompact) {
    __shared__ float2 hsum[UP == 2 ? 1024 : 1];
    const int bid = blockIdx.x;
    const int chunk = bid % nchunk, nc = bid / nchunk;
    const int n = nc / C, c = nc - n * C;
    const int gc = (n / Ng) * C + c;
    const int start = chunk * CHUNK;
    const int cnt = min(CHUNK, HW - start);
    const size_t off = (size_t)nc * HW + start;
    const size_t goff = ((size_t)n * 2 * C + c) * HW + start;
    const float mu = mean[gc], is = invstd[gc];
    float xh[16], g[16], t[16], acc[16];
    unsigned mask;
    load16_x<UP, true>(xs + (size_t)nc * (HW / (UP * UP)), start, cnt, W, xh);
#pragma unroll
    for (int i = 0; i < 16; ++i) { xh[i] = (xh[i] - mu) * is; acc[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* dy = k ? dy1 : dy0;
        const float* y = k ? y1 : y0;
        const float* gb = k ? gb1 : gb0;
        float* dgb = k ? dgb1 : dgb0;
        const float slope = k ? slope1 : slope0;
        const float m1 = sums[8 * gc + 4 * k] * inv_count, m2 = sums[8 * gc + 4 * k + 1] * inv_count;
        load16<true>(dy + off, cnt, g, mask);
        if (slope != 1.f) {
            load16<true>(y + off, cnt, t, mask);
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = t[i] > 0.f ? g[i] : g[i] * slope;
        }
        load16<true>(gb + (gcompact ? off : goff), cnt, t, mask);
        float dg[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dg[i] = g[i] * xh[i];
            acc[i] += g[i] * (1.f + t[i]) - m1 - xh[i] * m2;
        }
        if (dgb_split & 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { dg[i] = pack_split(dg[i]); g[i] = pack_split(g[i]); }
        }
        store16<true>(dgb + goff, cnt, dg);
        store16<true>(dgb + goff + (size_t)C * HW, cnt, g);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] *= is;
    if (UP == 1) {
        if (dgb_split & 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = pack_split(acc[i]);
        }
        store16<true>(dxs + off, cnt, acc);
        return;
    }
    // 2 x 2 reduction: each float4 holds two horizontal pairs; the vertical partner row is W / 4 float4 slots away (same chunk)
    const int wq = W >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = i * 256 + threadIdx.x;
        if (UP == 2) hsum[f] = make_float2(acc[4 * i] + acc[4 * i + 1], acc[4 * i + 2] + acc[4 * i + 3]);
    }
    __syncthreads();
    const int Ws = W >> 1;
    float* dplane = dxs + (size_t)nc * (HW >> 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = i * 256 + threadIdx.x, e = 4 * f;
        if (e >= cnt) continue;
        const int o = start + e;
        const int oy = o / W, ox = o - oy * W;
        if (oy & 1) continue;
        const float2 a = hsum[f], b = hsum[f + wq];
        float2 r = make_float2(a.x + b.x, a.y + b.y);
        if (dgb_split & 4) r = make_float2(pack_split(r.x), pack_split(r.y));
        *reinterpret_cast<float2*>(dplane + (size_t)(oy >> 1) * Ws + (ox >> 1)) = r;
    }
}

// ---------------------------------------------------------------- instance norm (+LeakyReLU): one workgroup per (n,c) plane
__global__ __launch_bounds__(256) void instnorm_lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                                 int HW, float eps, float slope) {
    __shared__ float sh[4];
    const size_t off = (size_t)blockIdx.x * HW;
    const float* xp = x + off;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
    s = block_sum_256(s, sh);
    const float mu = s / (float)HW;
    float m2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) { const float d = xp[i] - mu; m2 += d * d; }
    m2 = block_sum_256(m2, sh);
    const float is = 1.0f / sqrtf(m2 / (float)HW + eps);
    float* yp = y + off;
    for (int i = threadIdx.x; i < HW; i += 256) yp[i] = lrelu_f((xp[i] - mu) * is, slope);
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = mu; invstd_out[blockIdx.x] = is; }
}

__global__ __launch_bounds__(256) void instnorm_lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ dx,
                                                                 int HW, float slope) {
    __shared__ float sh[4];
    const size_t off = (size_t)blockIdx.x * HW;
    const float mu = mean[blockIdx.x], is = invstd[blockIdx.x];
    const float *dyp = dy + off, *yp = y + off, *xp = x + off;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float g = yp[i] > 0.f ? dyp[i] : dyp[i] * slope;
        s1 += g;
        s2 += g * ((xp[i] - mu) * is);
    }
    s1 = block_sum_256(s1, sh) / (float)HW;
    s2 = block_sum_256(s2, sh) / (float)HW;
    float* dxp = dx + off;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float g = yp[i] > 0.f ? dyp[i] : dyp[i] * slope;
        dxp[i] = is * (g - s1 - ((xp[i] - mu) * is) * s2);
    }
}

// ---------------------------------------------------------------- per-channel sum over (N, H*W): conv bias gradient
// grid (C, nsplit): split s sums images n = s, s+nsplit, ...; nsplit > 1 accumulates with one atomic per workgroup
// SPLIT: x holds split words; a value is hi + lo (exactly what the bf16x3 kernels multiply)
__device__ __forceinline__ float unsplit(float w) {
    const unsigned u = __float_as_uint(w);
    return __uint_as_float(u & 0xFFFF0000u) + __uint_as_float(u << 16);
}
template <bool SPLIT>
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                                          int HW, int nsplit, int acc) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int n = blockIdx.y; n < N; n += nsplit) {
        const float* p = x + ((size_t)n * C + c) * HW;
        if ((HW & 3) == 0 && ((((uintptr_t)p) & 15) == 0)) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
                float4 v = p4[i];
                if (SPLIT) { v.x = unsplit(v.x); v.y = unsplit(v.y); v.z = unsplit(v.z); v.w = unsplit(v.w); }
                s += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) s += SPLIT ? unsplit(p[i]) : p[i];
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) {
        if (nsplit > 1 || acc) atomicAdd(out + c, s); else out[c] = s;
    }
}

// ================================================================ C ABI
static inline bool vec_ok(int HW, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                          const void* e = nullptr, const void* f = nullptr, const void* g = nullptr) {
    auto al = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
    return (HW % 4 == 0) && al(a) && al(b) && al(c) && al(d) && al(e) && al(f) && al(g);
}

extern "C" size_t ag2v_bn_workspace_floats(int N, int C, int HW) {
    const int nchunk = ceil_div(HW, CHUNK);
    return (size_t)C * N * nchunk * 4 + 2 * (size_t)C;      // 3 floats per part for the statistics, 4 for the backward sums
}

// Batch statistics of x[N,C,H*W] per sample group of Ng consecutive samples (N % Ng == 0, G = N / Ng): mean / invstd [G, C]
// (biased variance + eps) and, when running_* are non-null, the G momentum updates of the running buffers in group order.
// workspace: ag2v_bn_workspace_floats() floats.
extern "C" int ag2v_bn_stats_g(const float* x, int N, int C, int HW, int Ng, float eps, float momentum, float* mean, float* invstd,
                               float* running_mean, float* running_var, float* workspace, void* stream) {
    if (!x || !mean || !invstd || !workspace || N <= 0 || C <= 0 || HW <= 0 || Ng <= 0 || N % Ng != 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    if (vec_ok(HW, x))
        hipLaunchKernelGGL(bn_stats_partial_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, st, x, workspace, Ng, C, HW, nchunk);
    else
        hipLaunchKernelGGL(bn_stats_partial_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, st, x, workspace, Ng, C, HW, nchunk);
    if (N / Ng > FIN_MAXG && running_mean) return AG2V_EINVAL;     // running-stat chain of more groups than the kernel stages
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, st, workspace, N / Ng, C, Ng * nchunk, eps,
                       momentum, mean, invstd, running_mean, running_var);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
// ag2v_norm_apply_gs(mode 0) that also writes the 2x nearest-neighbour down-sampling of y (y_half [N, C, H/2, W/2] = y[:, :, ::2, ::2],
// same word format) from the registers: the first level of the decoder's segmentation-map pyramid without a second pass over y.
// H % 2 == 0, W % 8 == 0, 16-byte aligned tensors.
extern "C" int ag2v_norm_apply_half(const float* x, const float* mean, const float* invstd, const float* w, const float* b, float* y,
                                    float* y_half, int N, int C, int H, int W, int Ng, float slope, int out_split, void* stream) {
    if (!x || !mean || !invstd || !y || !y_half || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 7) || Ng <= 0 || N % Ng != 0)
        return AG2V_EINVAL;
    const int HW = H * W;
    if (!vec_ok(HW, x, y) || (reinterpret_cast<uintptr_t>(y_half) & 15)) return AG2V_EINVAL;
    const int nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    hipLaunchKernelGGL((norm_apply_kernel<0, true>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, w, b,
                       (const float*)nullptr, (const float*)nullptr, y, C, HW, nchunk, slope, 0LL, Ng, out_split ? 1 : 0, y_half, W);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
// mean / invstd (and the running-statistics updates) of ag2v_bn_stats_g from partials another kernel wrote while it produced x
// (ag2v_layout_conv_img_fwd_stats): partial[G][C][nparts][3] = Chan's (count, mean, M2) of disjoint element sets covering each (group, channel).
extern "C" int ag2v_bn_finalize_partials(const float* partial, int G, int C, int nparts, float eps, float momentum, float* mean, float* invstd,
                                         float* running_mean, float* running_var, void* stream) {
    if (!partial || !mean || !invstd || G <= 0 || C <= 0 || nparts <= 0) return AG2V_EINVAL;
    if (G > FIN_MAXG && running_mean) return AG2V_EINVAL;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, G, C, nparts, eps,
                       momentum, mean, invstd, running_mean, running_var);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_bn_stats(const float* x, int N, int C, int HW, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, float* workspace, void* stream) {
    return ag2v_bn_stats_g(x, N, C, HW, N, eps, momentum, mean, invstd, running_mean, running_var, workspace, stream);
}

// mode 0: y = lrelu((x-mean)*invstd*w + b) (w/b may be null); mode 1: y = lrelu(xhat*(1+gamma)+beta). slope 1 => no activation.
// mean / invstd are [N / Ng, C]: sample n uses row n / Ng.
// out_split != 0: y is written as split words, (bf16(v) << 16) | bf16(v - bf16(v)) per value -- the form the bf16x3 convolution
// kernels otherwise compute from fp32 in their producers (ag2v_conv_opts::in_split).  A sign test on such a word (the backward
// pass reads y for the LeakyReLU branch) is a sign test on v.
extern "C" int ag2v_norm_apply_gs(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                                  const float* gamma, const float* beta, float* y, int N, int C, int HW, int Ng, float slope,
                                  int out_split, void* stream) {
    if (!x || !mean || !invstd || !y || N <= 0 || C <= 0 || HW <= 0 || Ng <= 0 || N % Ng != 0) return AG2V_EINVAL;
    if (mode < 0 || mode > 2 || (mode >= 1 && (!gamma || !beta))) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    const bool v = vec_ok(HW, x, y, gamma, beta);
    const long long gbx = (mode == 2) ? (long long)C * HW : 0;   // mode 2: gamma / beta are the halves of one [N,2C,HW] tensor
    dim3 g((unsigned)nblk), t(256);
    if (mode == 0) {
        if (v) hipLaunchKernelGGL((norm_apply_kernel<0, true>), g, t, 0, st, x, mean, invstd, w, b, gamma, beta, y, C, HW, nchunk, slope, gbx, Ng, out_split ? 1 : 0);
        else hipLaunchKernelGGL((norm_apply_kernel<0, false>), g, t, 0, st, x, mean, invstd, w, b, gamma, beta, y, C, HW, nchunk, slope, gbx, Ng, out_split ? 1 : 0);
    } else {
        if (v) hipLaunchKernelGGL((norm_apply_kernel<1, true>), g, t, 0, st, x, mean, invstd, w, b, gamma, beta, y, C, HW, nchunk, slope, gbx, Ng, out_split ? 1 : 0);
        else hipLaunchKernelGGL((norm_apply_kernel<1, false>), g, t, 0, st, x, mean, invstd, w, b, gamma, beta, y, C, HW, nchunk, slope, gbx, Ng, out_split ? 1 : 0);
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_norm_apply_g(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                                 const float* gamma, const float* beta, float* y, int N, int C, int HW, int Ng, float slope,
                                 void* stream) {
    return ag2v_norm_apply_gs(mode, x, mean, invstd, w, b, gamma, beta, y, N, C, HW, Ng, slope, 0, stream);
}
extern "C" int ag2v_norm_apply(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                               const float* gamma, const float* beta, float* y, int N, int C, int HW, float slope, void* stream) {
    return ag2v_norm_apply_g(mode, x, mean, invstd, w, b, gamma, beta, y, N, C, HW, N, slope, stream);
}

// Backward of ag2v_norm_apply_g with batch statistics (training mode), in two steps so that a data-parallel caller can all-reduce
// the per-(group, channel) sums between them (synchronised batch norm, sync_batchnorm/batchnorm.py:105-126):
//   reduce: sums[G*C*4] = (sum dxhat, sum dxhat*xhat, sum g, sum g*xhat) over this process's samples; mode 0: the first two are the
//           groups' (dbias, dweight); modes 1 / 2: the last two are the channel sums of (dbeta, dgamma)
//   apply:  dx (and, modes 1 / 2, dgamma / dbeta) from sums and inv_count = 1 / (#elements per (group, channel) the sums cover).
// y is the forward output (activation sign); may be null if slope == 1.
static int norm_bwd_launch(int mode, bool do_reduce, bool do_apply, const float* dy, const float* y, const float* x, const float* gamma,
                           const float* w, const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                           int N, int C, int HW, int Ng, float slope, float inv_count, float* workspace, hipStream_t st, int dgb_split = 0,
                           const float* bias = nullptr, bool sign_from_x = false) {
    if (!dy || !x || !mean || !invstd || !sums || N <= 0 || C <= 0 || HW <= 0 || Ng <= 0 || N % Ng != 0) return AG2V_EINVAL;
    if (do_reduce && !workspace) return AG2V_EINVAL;
    if (do_apply && !dx) return AG2V_EINVAL;
    if (slope != 1.f && !y && !(sign_from_x && mode == 0)) return AG2V_EINVAL;
    if (sign_from_x && mode == 0) y = nullptr;
    if (mode < 0 || mode > 3 || (mode >= 1 && !gamma) || (mode >= 1 && do_apply && (!dgamma || !dbeta))) return AG2V_EINVAL;
    const int nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    const bool v = vec_ok(HW, dy, y, x, gamma, dx, dgamma, dbeta);
    // mode 2: gamma, dgamma / dbeta are halves of [N,2C,HW] tensors; mode 3: the same gradients, gamma as a compact [N,C,HW] tensor
    const long long gbx = (mode >= 2) ? (long long)C * HW : 0;
    const long long ggx = (mode == 3) ? 0 : gbx;
    dim3 g((unsigned)nblk), t(256);
    const int GC = (N / Ng) * C;
    const int cpb = reduce_chunks_per_block(N * C, nchunk);
    const int nchunk_r = ceil_div(nchunk, cpb);
    dim3 gr((unsigned)((long long)N * C * nchunk_r));
#define LAUNCH_BWD(M, V)                                                                                                             \
    if (do_reduce) {                                                                                                                 \
        hipLaunchKernelGGL((norm_bwd_reduce_kernel<M, V>), gr, t, 0, st, dy, y, x, gamma, mean, invstd, workspace, Ng, C, HW, nchunk_r, \
                           slope, ggx, w, bias, cpb);                                                                                         \
        hipLaunchKernelGGL(pair_sum_finalize_kernel, dim3(ceil_div(GC, 64)), dim3(64), 0, st, workspace, GC, Ng* nchunk_r, sums);    \
    }                                                                                                                                \
    if (do_apply)                                                                                                                    \
        hipLaunchKernelGGL((norm_bwd_apply_kernel<M, V>), g, t, 0, st, dy, y, x, gamma, w, mean, invstd, sums, dx, dgamma, dbeta, C, \
                           HW, nchunk, inv_count, slope, gbx, Ng, dgb_split & 5, bias, ggx);
    if (mode == 0) {
        if (v) { LAUNCH_BWD(0, true) } else { LAUNCH_BWD(0, false) }
    } else {
        if (v) { LAUNCH_BWD(1, true) } else { LAUNCH_BWD(1, false) }
    }
#undef LAUNCH_BWD
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
// dgb_split (flag word), bit 0 (modes 1 / 2): dgamma / dbeta are written as split words (ag2v_conv_opts::dy_split) -- they are read by the
// data gradient and the weight gradient of the convolution that produced gamma | beta and by nothing else (its bias gradient comes
// from `sums`).  Bit 2 (value 4): dx is written as split words too -- for an x that one convolution produced and nothing else reads.
// mode 0 with y == NULL: the LeakyReLU branch of every element is recomputed from x, w and `bias` (the affine pair the forward
// pass applied: y = lrelu(xhat w + bias)) instead of being read from the forward output.
// _p ("phased"): phase 0 = both passes (ag2v_norm_bwd_gs); 1 = the reduce pass only (sums written); 2 = the apply pass only (sums read).
// Data-parallel SYNCHRONISED batch norm (sync_batchnorm/batchnorm.py:105-126) runs 1, all-reduces columns 0 / 1 of sums[N / Ng, C, 4]
// over the ranks, then 2 with inv_count = 1 / (elements per (group, channel) over ALL ranks); inv_count <= 0 = this process's own count.
extern "C" int ag2v_norm_bwd_gs_p(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                                  const float* bias, const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta,
                                  float* sums, int N, int C, int HW, int Ng, float slope, float* workspace, int dgb_split, int phase,
                                  float inv_count, void* stream) {
    if (Ng <= 0 || HW <= 0 || phase < 0 || phase > 2) return AG2V_EINVAL;
    if (!(inv_count > 0.f)) inv_count = 1.0f / ((float)Ng * (float)HW);
    return norm_bwd_launch(mode, phase != 2, phase != 1, dy, y, x, gamma, w, mean, invstd, dx, dgamma, dbeta, sums, N, C, HW, Ng, slope,
                           inv_count, workspace, (hipStream_t)stream, dgb_split, bias, mode == 0 && y == nullptr);
}
extern "C" int ag2v_norm_bwd_gs(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                                const float* bias, const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta,
                                float* sums, int N, int C, int HW, int Ng, float slope, float* workspace, int dgb_split, void* stream) {
    return ag2v_norm_bwd_gs_p(mode, dy, y, x, gamma, w, bias, mean, invstd, dx, dgamma, dbeta, sums, N, C, HW, Ng, slope, workspace,
                              dgb_split, 0, 0.f, stream);
}
// Backward of y = lrelu(batch_norm_affine(x)) (mode 0 of ag2v_norm_bwd_gs, the activation branch recomputed from x) whose output was
// read directly AND through nearest-neighbour down-samplings (ag2v_nearest_resize_fwd to hs[l] x ws[l]): the gradient
// g_full (nullable) + sum_l scatter(g_levels[l]) is formed while loading -- no materialised sum (ag2v_nearest_pyramid_bwd) in between.
// W % 4 == 0, 16-byte aligned tensors, H % hs[l] == 0, W % ws[l] == 0, at most 8 levels.
extern "C" int ag2v_norm_bwd_pyr_p(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                                   const float* x, const float* w, const float* bias, const float* mean, const float* invstd, float* dx,
                                   float* sums, int N, int C, int H, int W, int Ng, float slope, float* workspace, int phase,
                                   float inv_count_arg, void* stream) {
    if (!x || !mean || !invstd || !dx || !sums || !workspace || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (W & 3) || Ng <= 0 || N % Ng)
        return AG2V_EINVAL;
    if (phase < 0 || phase > 2) return AG2V_EINVAL;
    if (n_levels < 0 || n_levels > 8 || (n_levels && (!g_levels || !hs || !ws))) return AG2V_EINVAL;
    const int HW = H * W;
    if (!vec_ok(HW, g_full, x, dx)) return AG2V_EINVAL;
    NormPyramid lv{};
    lv.W = W;
    for (int l = 0; l < n_levels; ++l) {
        if (!g_levels[l]) continue;
        if (hs[l] <= 0 || ws[l] <= 0 || H % hs[l] || W % ws[l]) return AG2V_EINVAL;
        lv.g[lv.n] = g_levels[l]; lv.h[lv.n] = hs[l]; lv.w[lv.n] = ws[l]; lv.fy[lv.n] = H / hs[l]; lv.fx[lv.n] = W / ws[l];
        ++lv.n;
    }
    lv.pow2 = 1;
    for (int l = 0; l < lv.n; ++l)
        if ((lv.fy[l] & (lv.fy[l] - 1)) || (lv.fx[l] & (lv.fx[l] - 1))) lv.pow2 = 0;
    if (!g_full && lv.n == 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    dim3 g((unsigned)nblk), t(256);
    const int GC = (N / Ng) * C;
    const float inv_count = inv_count_arg > 0.f ? inv_count_arg : 1.0f / ((float)Ng * (float)HW);
    const int cpb = reduce_chunks_per_block(N * C, nchunk);
    const int nchunk_r = ceil_div(nchunk, cpb);
    if (phase != 2) {
        hipLaunchKernelGGL((norm_bwd_reduce_kernel<0, true, true>), dim3((unsigned)((long long)N * C * nchunk_r)), t, 0, st, g_full,
                           (const float*)nullptr, x, (const float*)nullptr, mean, invstd, workspace, Ng, C, HW, nchunk_r, slope, 0LL, w, bias, cpb, lv);
        hipLaunchKernelGGL(pair_sum_finalize_kernel, dim3(ceil_div(GC, 64)), dim3(64), 0, st, workspace, GC, Ng * nchunk_r, sums);
    }
    if (phase != 1)
        hipLaunchKernelGGL((norm_bwd_apply_kernel<0, true, true>), g, t, 0, st, g_full, (const float*)nullptr, x, (const float*)nullptr, w, mean,
                           invstd, sums, dx, (float*)nullptr, (float*)nullptr, C, HW, nchunk, inv_count, slope, 0LL, Ng, 0, bias, -1LL, lv);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_norm_bwd_pyr(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                                 const float* x, const float* w, const float* bias, const float* mean, const float* invstd, float* dx,
                                 float* sums, int N, int C, int H, int W, int Ng, float slope, float* workspace, void* stream) {
    return ag2v_norm_bwd_pyr_p(g_full, g_levels, hs, ws, n_levels, x, w, bias, mean, invstd, dx, sums, N, C, H, W, Ng, slope, workspace, 0,
                               0.f, stream);
}
extern "C" int ag2v_norm_bwd_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                               const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                               int N, int C, int HW, int Ng, float slope, float* workspace, void* stream) {
    return ag2v_norm_bwd_gs(mode, dy, y, x, gamma, w, nullptr, mean, invstd, dx, dgamma, dbeta, sums, N, C, HW, Ng, slope, workspace, 0, stream);
}
extern "C" int ag2v_norm_bwd_reduce_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* mean,
                                      const float* invstd, float* sums, int N, int C, int HW, int Ng, float slope, float* workspace,
                                      void* stream) {
    return norm_bwd_launch(mode, true, false, dy, y, x, gamma, nullptr, mean, invstd, nullptr, nullptr, nullptr, sums, N, C, HW, Ng, slope,
                           0.f, workspace, (hipStream_t)stream);
}
extern "C" int ag2v_norm_bwd_apply_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                                     const float* mean, const float* invstd, const float* sums, float inv_count, float* dx, float* dgamma,
                                     float* dbeta, int N, int C, int HW, int Ng, float slope, void* stream) {
    return norm_bwd_launch(mode, false, true, dy, y, x, gamma, w, mean, invstd, dx, dgamma, dbeta, const_cast<float*>(sums), N, C, HW, Ng,
                           slope, inv_count, nullptr, (hipStream_t)stream);
}
extern "C" int ag2v_norm_bwd(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                             const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                             int N, int C, int HW, float slope, float* workspace, void* stream) {
    return ag2v_norm_bwd_g(mode, dy, y, x, gamma, w, mean, invstd, dx, dgamma, dbeta, sums, N, C, HW, N, slope, workspace, stream);
}

// ---- SPADE pair (see the kernels): xs [N, C, H/up, W/up] (up = 1 or 2), gb0 / gb1 [N, 2C, H, W] -> y0 / y1 [N, C, H, W].
// Supported when W % 4 == 0, 16-byte aligned tensors and, for up = 2, 4096 % (2 W) == 0 and H, W even (ag2v_spade_pair_supported).
extern "C" int ag2v_spade_pair_supported(int H, int W, int up) {
    if (up != 1 && up != 2) return 0;
    if (W % 4 != 0 || ((long long)H * W) % 4 != 0) return 0;
    if (up == 2 && ((H & 1) || (W & 1) || CHUNK % (2 * W) != 0)) return 0;
    return 1;
}
extern "C" size_t ag2v_spade_pair_workspace_floats(int N, int C, int HW) {
    const int nchunk = ceil_div(HW, CHUNK);
    return (size_t)C * N * nchunk * 8;
}
// statistics of the (virtually upsampled) input: mean / invstd [N / Ng, C]; both running-stat pairs (nullable) updated per group
extern "C" int ag2v_spade_pair_stats(const float* xs, int N, int C, int HWs, int Ng, int rep, float eps, float momentum, float* mean,
                                     float* invstd, float* rm0, float* rv0, float* rm1, float* rv1, float* workspace, void* stream) {
    if (!xs || !mean || !invstd || !workspace || N <= 0 || C <= 0 || HWs <= 0 || Ng <= 0 || N % Ng != 0 || rep < 1) return AG2V_EINVAL;
    if ((rm0 == nullptr) != (rv0 == nullptr) || (rm1 == nullptr) != (rv1 == nullptr) || (rm1 && !rm0)) return AG2V_EINVAL;
    if (N / Ng > FIN_MAXG && rm0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = ceil_div(HWs, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    if (vec_ok(HWs, xs))
        hipLaunchKernelGGL(bn_stats_partial_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, st, xs, workspace, Ng, C, HWs, nchunk);
    else
        hipLaunchKernelGGL(bn_stats_partial_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, st, xs, workspace, Ng, C, HWs, nchunk);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, st, workspace, N / Ng, C, Ng * nchunk, eps, momentum,
                       mean, invstd, rm0, rv0, rep, rm1, rv1);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
// split0 / split1 != 0: y0 / y1 stored as split words (see ag2v_norm_apply_gs)
extern "C" int ag2v_spade_pair_apply_s(const float* xs, const float* mean, const float* invstd, const float* gb0, const float* gb1, float* y0,
                                       float* y1, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, int split0,
                                       int split1, void* stream) {
    if (!xs || !mean || !invstd || !gb0 || !gb1 || !y0 || !y1 || N <= 0 || C <= 0 || Ng <= 0 || N % Ng != 0) return AG2V_EINVAL;
    if (!ag2v_spade_pair_supported(H, W, up) || !vec_ok(H * W, xs, gb0, gb1, y0, y1)) return AG2V_EINVAL;
    const int HW = H * W, nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    dim3 g((unsigned)nblk), t(256);
    if (up == 1) hipLaunchKernelGGL(spade_pair_apply_kernel<1>, g, t, 0, (hipStream_t)stream, xs, mean, invstd, gb0, gb1, y0, y1, C, HW, W, nchunk, slope0, slope1, Ng, split0 ? 1 : 0, split1 ? 1 : 0);
    else hipLaunchKernelGGL(spade_pair_apply_kernel<2>, g, t, 0, (hipStream_t)stream, xs, mean, invstd, gb0, gb1, y0, y1, C, HW, W, nchunk, slope0, slope1, Ng, split0 ? 1 : 0, split1 ? 1 : 0);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_spade_pair_apply(const float* xs, const float* mean, const float* invstd, const float* gb0, const float* gb1, float* y0,
                                     float* y1, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, void* stream) {
    return ag2v_spade_pair_apply_s(xs, mean, invstd, gb0, gb1, y0, y1, N, C, H, W, up, Ng, slope0, slope1, 0, 0, stream);
}
// sums [N / Ng, C, 8] = per norm k: (sum dxhat_k, sum dxhat_k xhat, sum dbeta_k, sum dgamma_k); dxs [N, C, H/up, W/up]; dgb0 / dgb1 like gb.
// _sp: phases as ag2v_norm_bwd_gs_p (synchronised batch norm all-reduces columns 0, 1, 4, 5 of sums[N / Ng, C, 8] between them)
extern "C" int ag2v_spade_pair_bwd_sp(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                                    const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1,
                                    float* sums, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace,
                                    int dgb_split, int phase, float inv_count_arg, void* stream) {
    if (phase < 0 || phase > 2) return AG2V_EINVAL;
    if (!dy0 || !dy1 || !xs || !gb0 || !gb1 || !mean || !invstd || !dxs || !dgb0 || !dgb1 || !sums || !workspace) return AG2V_EINVAL;
    if (N <= 0 || C <= 0 || Ng <= 0 || N % Ng != 0 || (slope0 != 1.f && !y0) || (slope1 != 1.f && !y1)) return AG2V_EINVAL;
    if (!ag2v_spade_pair_supported(H, W, up) || !vec_ok(H * W, dy0, dy1, y0, y1, gb0, gb1) || !vec_ok(H * W, dgb0, dgb1, xs, dxs)) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // dgb_split is a flag word: 1 = dgb0 / dgb1 written as split words; 2 = gb0 / gb1 point at compact gamma copies [N, C, H, W]
    // (the backward never reads beta: a caller that keeps only gamma frees half of every gamma | beta tensor of the tape)
    // 4 = dxs written as split words (xs produced by one convolution and read by this pair only)
    const int gcompact = (dgb_split >> 1) & 1;
    dgb_split &= 5;
    const int HW = H * W, nchunk = ceil_div(HW, CHUNK);
    const long long nblk = (long long)N * C * nchunk;
    if (nblk >= (1LL << 31)) return AG2V_EINVAL;
    dim3 g((unsigned)nblk), t(256);
    const int GC = (N / Ng) * C;
    const float inv_count = inv_count_arg > 0.f ? inv_count_arg : 1.0f / ((float)Ng * (float)HW);
    const int cpb = reduce_chunks_per_block(N * C, nchunk);
    const int nchunk_r = ceil_div(nchunk, cpb);
    dim3 gr((unsigned)((long long)N * C * nchunk_r));
#define PAIR_BWD(U)                                                                                                                  \
    if (phase != 2) {                                                                                                                \
        hipLaunchKernelGGL(spade_pair_bwd_reduce_kernel<U>, gr, t, 0, st, dy0, y0, dy1, y1, xs, gb0, gb1, mean, invstd, workspace, Ng, C, HW, \
                           W, nchunk_r, slope0, slope1, gcompact, cpb);                                                              \
        hipLaunchKernelGGL(multi_sum_finalize_kernel, dim3(ceil_div(GC * 8, 64)), dim3(64), 0, st, workspace, GC, Ng* nchunk_r, 8, sums); \
    }                                                                                                                                \
    if (phase != 1)                                                                                                                  \
        hipLaunchKernelGGL(spade_pair_bwd_apply_kernel<U>, g, t, 0, st, dy0, y0, dy1, y1, xs, gb0, gb1, mean, invstd, sums, dxs, dgb0, dgb1, \
                           C, HW, W, nchunk, inv_count, slope0, slope1, Ng, dgb_split, gcompact);
    if (up == 1) { PAIR_BWD(1) } else { PAIR_BWD(2) }
#undef PAIR_BWD
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_spade_pair_bwd_s(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                                   const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1,
                                   float* sums, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace,
                                   int dgb_split, void* stream) {
    return ag2v_spade_pair_bwd_sp(dy0, y0, dy1, y1, xs, gb0, gb1, mean, invstd, dxs, dgb0, dgb1, sums, N, C, H, W, up, Ng, slope0, slope1,
                                  workspace, dgb_split, 0, 0.f, stream);
}
extern "C" int ag2v_spade_pair_bwd(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                                   const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1,
                                   float* sums, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace,
                                   void* stream) {
    return ag2v_spade_pair_bwd_s(dy0, y0, dy1, y1, xs, gb0, gb1, mean, invstd, dxs, dgb0, dgb1, sums, N, C, H, W, up, Ng, slope0, slope1,
                                 workspace, 0, stream);
}

extern "C" int ag2v_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* invstd, int NC, int HW, float eps,
                                       float slope, void* stream) {
    if (!x || !y || !mean || !invstd || NC <= 0 || HW <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(instnorm_lrelu_fwd_kernel, dim3(NC), dim3(256), 0, (hipStream_t)stream, x, y, mean, invstd, HW, eps, slope);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_instnorm_lrelu_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                       float* dx, int NC, int HW, float slope, void* stream) {
    if (!dy || !y || !x || !mean || !invstd || !dx || NC <= 0 || HW <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(instnorm_lrelu_bwd_kernel, dim3(NC), dim3(256), 0, (hipStream_t)stream, dy, y, x, mean, invstd, dx, HW, slope);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// accumulate != 0: out[c] += sum (a bias used by several forward calls of one step collects its gradient in one buffer)
// x_split != 0: x holds split words (a gradient written by a convolution epilogue with ag2v_conv_opts::out_split); the sum is over hi + lo
// accumulate is a flag word: AG2V_ACCUMULATE (1) | AG2V_DETERMINISTIC (2: one workgroup per channel sums every image in order -- no
// floating-point atomics between image splits; an accumulating launch then adds its one sum per channel, which is order-free)
extern "C" int ag2v_channel_sum_acc_s(const float* x, float* out, int N, int C, int HW, int accumulate, int x_split, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool det = (accumulate & 2) != 0;
    accumulate &= 1;
    int nsplit = 1;
    if ((long long)C < 1024 && (long long)HW * N >= (1 << 16) && !det) {
        nsplit = ceil_div(1024, C);
        if (nsplit > N) nsplit = N;
    }
    if (nsplit > 1 && !accumulate && hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    if (x_split) hipLaunchKernelGGL(channel_sum_kernel<true>, dim3(C, nsplit), dim3(256), 0, st, x, out, N, C, HW, nsplit, accumulate ? 1 : 0);
    else hipLaunchKernelGGL(channel_sum_kernel<false>, dim3(C, nsplit), dim3(256), 0, st, x, out, N, C, HW, nsplit, accumulate ? 1 : 0);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_channel_sum_acc(const float* x, float* out, int N, int C, int HW, int accumulate, void* stream) {
    return ag2v_channel_sum_acc_s(x, out, N, C, HW, accumulate, 0, stream);
}
extern "C" int ag2v_channel_sum(const float* x, float* out, int N, int C, int HW, void* stream) {
    return ag2v_channel_sum_acc(x, out, N, C, HW, 0, stream);
}
