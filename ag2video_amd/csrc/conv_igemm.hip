// Implicit-GEMM 2-D convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 157.3 TF peak).
//
// Replaces every nn.Conv2d call site of the reference hot path:
//   generator 3x3 / 1x1 convs   models/spade_models/networks/{generator.py:30, flows_generator.py:32-54,82-85,
//                               spade_generator.py:16-34, architecture.py:29-32, normalization.py:90-94}
//   discriminator 4x4 s2/s1 p2  models/spade_models/networks/discriminator.py:368-379
// forward, data gradient and weight gradient (what autograd derives from aten.convolution there).
//
// One forward kernel serves the forward pass AND every data gradient:
//   FWD    Y[k,(n,oy,ox)] = sum_{c,r,s} At[(c,r,s),k] * X[n,c,oy*st+r-pad,ox*st+s-pad]          At = W^T      (prep kernel)
//   DGRAD  for each output-pixel parity class (py,px) of the stride:  dX[c, class pixels] =
//          conv_stride1(dY, sub-filter r = py+st*r', s = px+st*s', flipped)                      At = sub-filter^T (prep kernel)
//          (stride 1: one class = the classic flipped-filter identity; stride 2: four classes with 1/4 of the taps each,
//           instead of multiplying 3/4 structural zeros)
//   WGRAD  dW[k,(c,r,s)]  = sum_{n,oy,ox} dY[n,k,oy,ox] * X[n,c,oy*st+r-pad,ox*st+s-pad]         (own kernel, split over pixels)
// Both operands reach LDS through fully coalesced loads (lanes along m for the k-major weights, lanes along the pixel index
// for the activation gather), are staged as [k][m] / [k][n] so every MFMA operand read is a conflict-free ds_read_b32, and
// the next K-tile's global loads are in flight while the current tile's 32 MFMAs per wave issue.
#include "conv_common.h"
#include <stdlib.h>
#include <string.h>

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, (BM == 64 ? 3 : 4)) void conv_igemm_kernel(ConvArgs p) {   // <64,256>: 3 resident (VGPRs)
    constexpr int BK = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int EA = BM * BK / 256, KSA = 256 / BM;   // A: one m column, EA k-rows (step KSA) per thread
    constexpr int EB = BK * BN / 256, KSB = 256 / BN;   // B: one pixel column, EB k-rows (step KSB) per thread
    static_assert(TM >= 1 && TN >= 1, "tile");
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
    // K-index decode table: the 16 k-rows of a K-tile are decoded ONCE per workgroup (16 lanes in parallel) into LDS as
    // {offset of (chan,tap) inside a source image, tap bit}; every gathered element then costs one broadcast ds_read_b64,
    // a mask test and one 64-bit add.  (A per-element scalar decode cost ~10 SALU instructions per MFMA.)
    __shared__ int2 tab[2][BK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;  // m-tiles of one pixel tile are neighbours
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
    const int RS = p.R * p.S;
    const int SHW = p.H * p.W;

    // ---- B gather state: this thread owns pixel column jb of the tile for the whole K loop
    const int jb = tid % BN, kkb0 = tid / BN;
    const int pix = n0 + jb;
    const bool pvalid = pix < p.NG;
    const int n_img = pix / (p.OH * p.OW);
    const int prem = pix - n_img * (p.OH * p.OW);
    const int py = prem / p.OW, px = prem - py * p.OW;
    const int y0 = py * p.stride - p.pad, x0 = px * p.stride - p.pad_w;
    // bit (r*S+s) of vmask: tap (r,s) of this pixel reads inside the source plane; its address is bptr + chan*SHW + r*W + s
    unsigned vmask = 0;
    if (pvalid) {
        for (int r = 0; r < p.R; ++r)
            for (int s2 = 0; s2 < p.S; ++s2)
                if ((unsigned)(y0 + r) < (unsigned)p.H && (unsigned)(x0 + s2) < (unsigned)p.W) vmask |= 1u << (r * p.S + s2);
    }
    const float* bptr = p.b_src + (size_t)n_img * p.C * SHW + (y0 * p.W + x0);  // tap (0,0); may point before the plane
    // ---- A load state: k-major weights, lanes along m
    const int am = tid % BM, akk0 = tid / BM;
    const bool mvalid = (m0 + am) < p.M;
    const float* aptr = p.a_src + m0 + am;

    auto build_tab = [&](int kt) {
        if (tid < BK) {
            const int kg = kt * BK + tid;
            int c, r, s2;
            decode_crs(kg, RS, p.S, c, r, s2);
            const unsigned bit = (kg < p.KG) ? (1u << (r * p.S + s2)) : 0u;
            tab[kt & 1][tid] = make_int2(c * SHW + r * p.W + s2, (int)bit);
        }
    };
    auto fetchB = [&](int kt, int kk) -> float {
        const int2 e = tab[kt & 1][kk];
        return (vmask & (unsigned)e.y) ? bptr[e.x] : 0.f;
    };
    auto fetchA = [&](int kg) -> float { return (mvalid && kg < p.KG) ? aptr[(size_t)kg * p.M] : 0.f; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[EA], rb[EB];
    const int nk_all = (p.KG + BK - 1) / BK;
    const int kt0 = (p.splitk > 1) ? blockIdx.y * p.kchunk : 0;  // small-N layers: the K loop is split over grid.y
    const int nk = (p.splitk > 1) ? min(nk_all, kt0 + p.kchunk) : nk_all;
    if (kt0 >= nk) return;  // uniform per workgroup
    build_tab(kt0);
    __syncthreads();
    build_tab(kt0 + 1);
#pragma unroll
    for (int i = 0; i < EA; ++i) ra[i] = fetchA(kt0 * BK + akk0 + KSA * i);
#pragma unroll
    for (int i = 0; i < EB; ++i) rb[i] = fetchB(kt0, kkb0 + KSB * i);
#pragma unroll
    for (int i = 0; i < EA; ++i) As[0][akk0 + KSA * i][am] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[0][kkb0 + KSB * i][jb] = rb[i];
    __syncthreads();

    int cur = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            const int k0 = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < EA; ++i) ra[i] = fetchA(k0 + akk0 + KSA * i);
#pragma unroll
            for (int i = 0; i < EB; ++i) rb[i] = fetchB(kt + 1, kkb0 + KSB * i);
            build_tab(kt + 2);  // tab[kt&1] was last read while fetching tile kt, i.e. before the previous barrier
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][kk][wm0 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][kk][wn0 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < EA; ++i) As[cur ^ 1][akk0 + KSA * i][am] = ra[i];
#pragma unroll
            for (int i = 0; i < EB; ++i) Bs[cur ^ 1][kkb0 + KSB * i][jb] = rb[i];
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int gplane = p.OH * p.OW;   // pixels of the GEMM's pixel space per image
    const int plane = p.o_H * p.o_W;  // elements of one output channel plane
    const bool mapped = (p.o_mul != 1 || p.o_add_y != 0 || p.o_add_x != 0 || p.o_W != p.OW);
    // Reads of the epilogue (bias, residual or mask word) are requested for 8 registers at a time and waited for once; the
    // arithmetic on them is branch-free (uniform multipliers / thresholds).  Row by row, every conditional load was followed by
    // s_waitcnt vmcnt(0) -- stores included -- i.e. up to three serialised memory round trips per accumulator register.
    const float ninf = -__builtin_huge_valf();
    const float rmul = (p.residual && kt0 == 0) ? 1.f : 0.f;
    const float relu_floor = (p.relu && p.splitk == 1) ? 0.f : ninf;
    const float mask_thr = (p.mask && p.splitk == 1) ? 0.f : ninf;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int opix = n0 + wn0 + j * 32 + (lane & 31);
        bool pvalid = opix < p.NG;
        const int on = pvalid ? opix / gplane : 0;
        int orem = pvalid ? opix - on * gplane : 0;
        if (mapped) {
            const int qy = orem / p.OW, qx = orem - qy * p.OW;
            const int yy = qy * p.o_mul + p.o_add_y, xx = qx * p.o_mul + p.o_add_x;
            pvalid = pvalid && (unsigned)yy < (unsigned)p.o_H && (unsigned)xx < (unsigned)p.o_W;
            orem = pvalid ? yy * p.o_W + xx : 0;
        }
        float* obase = p.out + (size_t)on * p.out_nstride + orem;
        const float* rbase = p.residual ? p.residual + (size_t)on * p.M * plane + orem : nullptr;
        const float* mbase = (p.mask && p.splitk == 1) ? p.mask + (size_t)on * p.M * plane + orem : nullptr;
        const float* aux = rbase ? rbase : mbase;           // at most one of them (ag2v_conv2d_dgrad_acc rejects both)
        const bool want_bias = p.bias && kt0 == 0;
        const float gs = p.gscale ? p.gscale[on / p.group_n] : 1.f;   // per-sample-group scale of the product (1/sigma_t)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float b[8], ax[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { b[k] = 0.f; ax[k] = 0.f; }
                if (want_bias || aux) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = 8 * h + k;
                        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (m < p.M) {
                            if (want_bias) b[k] = p.bias[m];
                            if (aux && pvalid) ax[k] = aux[(size_t)m * plane];
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), once per 8 registers
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float x = fmaf(acc[i][j][8 * h + k], gs, b[k]);
                    x = fmaf(ax[k], rmul, x);
                    x = fmaxf(x, relu_floor);
                    b[k] = (ax[k] > mask_thr) ? x : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = 8 * h + k;
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (pvalid && m < p.M) {
                        if (p.splitk > 1)
                            atomicAdd(obase + (size_t)m * plane, b[k]);
                        else
                            obase[(size_t)m * plane] = b[k];
                    }
                }
            }
        }
    }
}

// Weight gradient: M = K (rows from dY), columns j = (c,r,s) gathered from X, reduction over pixels q=(n,oy,ox).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvArgs p) {
    constexpr int BK = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int EA = BM / 16, EB = BN / 16;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
    const int RS = p.R * p.S;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    // pixel range of this workgroup: split sl of sample group grp (groups == 1: the whole batch is one group)
    const int grp = blockIdx.y / p.spg, sl = blockIdx.y - grp * p.spg;
    const int qs = grp * p.gq + sl * p.kchunk;
    const int qe = min(qs + p.kchunk, (grp + 1) * p.gq);
    if (qs >= qe) return;  // uniform per block
    float* const outg = p.det ? p.out + (size_t)blockIdx.y * ((size_t)p.M * p.NG) : p.out + (size_t)grp * p.out_gstride;
    const bool atomic_out = !p.det && (p.spg > 1 || p.accumulate);

    const int qq = tid & 15, rg = tid >> 4;
    // hoisted column decode: j -> (c, r-pad, s-pad)
    int boff[EB], bpk[EB];   // bpk: (r-pad+64) | (s-pad+64)<<8 | valid<<16
#pragma unroll
    for (int i = 0; i < EB; ++i) {
        const int j = n0 + rg + 16 * i;
        int c, r, s;
        decode_crs(j, RS, p.S, c, r, s);
        bpk[i] = ((r - p.pad + 64) & 255) | (((s - p.pad + 64) & 255) << 8) | ((j < p.NG) ? (1 << 16) : 0);
        boff[i] = c * HW + (r - p.pad) * p.W + (s - p.pad);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[EA], rb[EB];
    auto fetch = [&](int q) {
        const bool qvalid = q < qe;
        const int qc = qvalid ? q : qs;
        const int n = qc / OHW;
        const int pixl = qc - n * OHW;
        const int oy = pixl / p.OW, ox = pixl - oy * p.OW;
        const int yb = oy * p.stride, xb = ox * p.stride;
        const float* ap = p.a_src + (size_t)n * p.K * OHW + pixl;
        const float* bp = p.b_src + (size_t)n * p.C * HW + yb * p.W + xb;
#pragma unroll
        for (int i = 0; i < EA; ++i) {
            const int m = m0 + rg + 16 * i;
            ra[i] = (qvalid && m < p.M) ? ap[(size_t)m * OHW] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int iy = yb + (bpk[i] & 255) - 64, ix = xb + ((bpk[i] >> 8) & 255) - 64;
            const bool ok = qvalid && (bpk[i] >> 16) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            rb[i] = ok ? bp[boff[i]] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < EA; ++i) As[buf][qq][rg + 16 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < EB; ++i) Bs[buf][qq][rg + 16 * i] = rb[i];
    };

    const int nk = (qe - qs + BK - 1) / BK;
    fetch(qs + qq);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) fetch(qs + (kt + 1) * BK + qq);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][kk][wm0 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][kk][wn0 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + (lane & 31);
        if (col >= p.NG) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float* dst = outg + (size_t)m * p.NG + col;
                    if (atomic_out)
                        atomicAdd(dst, acc[i][j][r]);
                    else
                        *dst = acc[i][j][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight preparation (k-major A operands)
// at[col][row] = w[row][col]  (w: [rows][cols] row-major): 32x32 LDS tiles, coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ w, float* __restrict__ at, int rows, int cols,
                                                        int split) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) {
            const float v = t[tx][ty + 8 * i];
            at[(size_t)c * rows + r] = (split == 2) ? pack_split_h(v) : (split ? pack_split(v) : v);      // 2: fp16 planes (ConvArgs::h16)
        }
    }
}

// Forward weights in the tap-major layout of the bf16x3 implicit GEMM (ConvArgs::tap_major), C % 32 == 0: two bf16 planes
// (hi = bf16(w), lo = bf16(w - hi)), the lo plane KG * M bf16 behind the hi plane, each [KG/8][M][8]:
//   plane[((tap_major_row(c, r, s) >> 3) * M + m) * 8 + (c & 7)] = hi | lo of w[m][c][r][s]
// grid (C/32, ceil(M/8)): the 32*R*S contiguous source floats of each of 8 output rows go through LDS; every thread then writes
// the eight channels c..c+7 of one (tap, m) as 16 bytes per plane, 8 lanes along m (one 128-byte line).  Small blocks: the
// weights of a 128 x 512 x 3 x 3 layer make 256 of them, and the kernel runs in front of every forward convolution.
__device__ __forceinline__ void split_planes8(const float (&x)[8], uint4& hi, uint4& lo, int h16 = 0) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = __float_as_uint(h16 ? pack_split_h(x[2 * i]) : pack_split(x[2 * i]));
        const unsigned b = __float_as_uint(h16 ? pack_split_h(x[2 * i + 1]) : pack_split(x[2 * i + 1]));
        h[i] = (a >> 16) | (b & 0xFFFF0000u);
        l[i] = (a & 0xFFFFu) | (b << 16);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__global__ __launch_bounds__(256) void pack_fwd_tap_kernel(const float* __restrict__ w, float* __restrict__ at, int M, int C, int R,
                                                           int S, int h16) {
    constexpr int MR = 8;
    __shared__ float t[MR * (32 * 16 + 1)];
    const int RS = R * S, row = 32 * RS + 1;
    const int c0 = blockIdx.x * 32, m0 = blockIdx.y * MR;
    for (int i = threadIdx.x; i < MR * 32 * RS; i += 256) {
        const int m = i / (32 * RS), j = i - m * (32 * RS);
        t[m * row + j] = (m0 + m < M) ? w[((size_t)(m0 + m) * C + c0) * RS + j] : 0.f;
    }
    __syncthreads();
    uint4* hi_plane = reinterpret_cast<uint4*>(at);
    uint4* lo_plane = hi_plane + (size_t)(C * RS / 8) * M;
    for (int i = threadIdx.x; i < RS * 4 * MR; i += 256) {
        const int m = i & (MR - 1), c8 = (i >> 3) & 3, tap = i >> 5;
        if (m0 + m >= M) continue;
        const int r = tap / S, s2 = tap - r * S;
        const float* src = &t[m * row + (c8 * 8) * RS + tap];
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = src[e * RS];
        uint4 hi, lo;
        split_planes8(x, hi, lo, h16);
        const size_t idx = (size_t)(tap_major_row(c0 + c8 * 8, r, s2, R, S) >> 3) * M + (m0 + m);
        hi_plane[idx] = hi;
        lo_plane[idx] = lo;
    }
}

// Data-gradient sub-filter of pixel class (py,px), flipped, k-major:
//   at[((k*Rc + r2)*Sc + s2)*C + c] = w[k][c][py + st*(Rc-1-r2)][px + st*(Sc-1-s2)]
// grid (ceil(C/64), K): the 64*R*S contiguous source floats of (k, c-chunk) go through LDS, rows are written coalesced.
__global__ __launch_bounds__(256) void dgrad_weight_prep_kernel(const float* __restrict__ w, float* __restrict__ at, int K, int C,
                                                                int R, int S, int py, int px, int st, int Rc, int Sc, int split) {
    __shared__ float t[64 * 16];
    const int RS = R * S;
    const int c0 = blockIdx.x * 64, k = blockIdx.y;
    const int nc = min(64, C - c0);
    const float* src = w + ((size_t)k * C + c0) * RS;
    for (int i = threadIdx.x; i < nc * RS; i += 256) t[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int tap = grp; tap < Rc * Sc; tap += 4) {
        const int r2 = tap / Sc, s2 = tap - r2 * Sc;
        const int r = py + st * (Rc - 1 - r2), s = px + st * (Sc - 1 - s2);
        if (lane < nc) {
            const float v = t[lane * RS + r * S + s];
            at[((size_t)(k * Rc + r2) * Sc + s2) * C + c0 + lane] = split ? pack_split(v) : v;
        }
    }
}

// The same sub-filter in the tap-major plane layout (ConvArgs::tap_major, see pack_fwd_tap_kernel; the gathered channels of the
// data-gradient GEMM are the K forward outputs, its rows the C forward inputs), K % 32 == 0:
//   plane[((tap_major_row(k, r2, s2) >> 3) * C + c) * 8 + (k & 7)] = hi | lo of w[k][c][py + st*(Rc-1-r2)][px + st*(Sc-1-s2)]
// grid (ceil(C/64), K/8): eight k per block, so that every thread writes 16 bytes per plane, lanes along c.
__global__ __launch_bounds__(256) void pack_dgrad_tap_kernel(const float* __restrict__ w, float* __restrict__ at, int K, int C, int R,
                                                             int S, int py, int px, int st, int Rc, int Sc) {
    __shared__ float t[8][64 * 17];
    const int RS = R * S, RSP = RS | 1;      // odd row length: lanes (stride RSP) on distinct banks
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 8;
    const int nc = min(64, C - c0);
    for (int j = 0; j < 8; ++j) {
        const float* src = w + ((size_t)(k0 + j) * C + c0) * RS;
        for (int i = threadIdx.x; i < nc * RS; i += 256) {
            const int c = i / RS;
            t[j][c * RSP + (i - c * RS)] = src[i];
        }
    }
    __syncthreads();
    uint4* hi_plane = reinterpret_cast<uint4*>(at);
    uint4* lo_plane = hi_plane + (size_t)(K * Rc * Sc / 8) * C;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int tap = grp; tap < Rc * Sc; tap += 4) {
        const int r2 = tap / Sc, s2 = tap - r2 * Sc;
        const int q = lane * RSP + (py + st * (Rc - 1 - r2)) * S + (px + st * (Sc - 1 - s2));
        if (lane < nc) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = t[e][q];
            uint4 hi, lo;
            split_planes8(x, hi, lo);
            const size_t idx = (size_t)(tap_major_row(k0, r2, s2, Rc, Sc) >> 3) * C + (c0 + lane);
            hi_plane[idx] = hi;
            lo_plane[idx] = lo;
        }
    }
}

static thread_local const char* g_last_conv_kernel = "";
void set_last_conv_kernel(const char* name) { g_last_conv_kernel = name; }
// (ag2v_conv_opts::chan_sums_done: set by the launcher whose kernel adds the channel sums in its epilogue -- conv_halo.hip)
static thread_local int g_chan_sums_done = 0;
void note_chan_sums_done(int v) { g_chan_sums_done = v; }
static int chan_sums_done() { return g_chan_sums_done; }
extern "C" const char* ag2v_last_conv_kernel(void) { return g_last_conv_kernel; }

static int conv_out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

// Arithmetic of the 128/256-row igemm / wgrad tiles: 1 = bf16x3 split (fp32-level accuracy, ~1e-5 relative per product, at up to
// 16/3 of the fp32 MFMA rate), 0 = exact fp32 MFMA (bit-for-bit fmaf chains); GEMMs with fewer rows than min_rows stay exact.
// Both are PER-CALL arguments (ag2v_conv_opts of the *_ex entry points): a call that names them reads nothing but its arguments.
// A call that does NOT name them (no options, precision < 0, min_rows <= 0) takes the defaults of the DEVICE it runs on: a per-device
// record (no global mutable state beyond a per-device handle), initialised once from AG2V_CONV_PRECISION=fp32|bf16x3 /
// AG2V_BF16X3_MIN_M and changed only by the deprecated ag2v_set_* shims for the calling thread's current device.  The Python host
// never relies on them: ops.py passes both fields with every call.
#include <atomic>
static constexpr int kMaxDevices = 64;
struct DeviceDefaults { std::atomic<int> precision{-1}, min_rows{-1}; };
static DeviceDefaults g_dev_defaults[kMaxDevices];
static DeviceDefaults& device_defaults() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return g_dev_defaults[dev];
}
static int conv_precision() {
    DeviceDefaults& d = device_defaults();
    int v = d.precision.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("AG2V_CONV_PRECISION");
        v = (e && !strcmp(e, "fp32")) ? 0 : 1;
        d.precision.store(v, std::memory_order_relaxed);
    }
    return v;
}
/* deprecated: pass ag2v_conv_opts::precision */
extern "C" int ag2v_set_conv_precision(int mode) {
    if (mode != 0 && mode != 1) return AG2V_EINVAL;
    device_defaults().precision.store(mode, std::memory_order_relaxed);
    return AG2V_OK;
}
extern "C" int ag2v_get_conv_precision(void) { return conv_precision(); }
static int bf16x3_min_rows() {
    DeviceDefaults& d = device_defaults();
    int v = d.min_rows.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("AG2V_BF16X3_MIN_M");
        v = (e && atoi(e) > 0) ? atoi(e) : 128;
        d.min_rows.store(v, std::memory_order_relaxed);
    }
    return v;
}
/* deprecated: pass ag2v_conv_opts::min_rows */
extern "C" int ag2v_set_bf16x3_min_rows(int rows) {
    if (rows < 1) return AG2V_EINVAL;
    device_defaults().min_rows.store(rows, std::memory_order_relaxed);
    return AG2V_OK;
}
extern "C" int ag2v_get_bf16x3_min_rows(void) { return bf16x3_min_rows(); }

// mirror of ag2v_conv_opts (include/ag2v.h)
struct ag2v_conv_opts {
    int precision;
    int min_rows;
    const float* gscale;
    int group_n;
    const char* kernel;
    int in_split;
    int dy_split;
    int out_split;
    int deterministic;
    float* det_workspace;
    size_t det_workspace_floats;
    long long out_nstride;
    long long out_mstride;
    float* out_chan_sums;
    int chan_sums_done;
};
struct ConvCfg {
    int precision, min_rows;
    const float* gscale;
    int group_n;
    int in_split, dy_split, out_split;
    int det;
    float* det_ws;
    size_t det_ws_floats;
    long long out_nstride;
    long long out_mstride;
    float* chan_sums;
};
static ConvCfg conv_cfg(const ag2v_conv_opts* o) {
    ConvCfg c;
    c.precision = (o && o->precision >= 0) ? (o->precision == 2 ? 2 : (o->precision ? 1 : 0)) : conv_precision();
    c.min_rows = (o && o->min_rows > 0) ? o->min_rows : bf16x3_min_rows();
    c.gscale = o ? o->gscale : nullptr;
    c.group_n = (o && o->group_n > 0) ? o->group_n : 1;
    c.in_split = (o && o->in_split) ? 1 : 0;
    c.dy_split = (o && o->dy_split) ? 1 : 0;
    c.out_split = (o && o->out_split) ? 1 : 0;
    c.det = (o && o->deterministic) ? 1 : 0;
    c.det_ws = o ? o->det_workspace : nullptr;
    c.det_ws_floats = o ? o->det_workspace_floats : 0;
    c.out_nstride = o ? o->out_nstride : 0;
    c.out_mstride = o ? o->out_mstride : 0;
    c.chan_sums = o ? o->out_chan_sums : nullptr;
    return c;
}
static bool use_bf16x3(const ConvCfg& c, int M) { return c.precision >= 1 && M >= c.min_rows; }     // (2: fp16 split in the forward pass)
static const long long kBufMax = (1LL << 32) - 64;   // raw buffer descriptors cover < 4 GiB


// ---------------------------------------------------------------- ordered reduction of a deterministic weight-gradient launch
// (ConvArgs::det): dw[g][e] = (acc ? dw[g][e] : 0) + slab[g * spg + 0][e] + slab[g * spg + 1][e] + ...  -- the pixel-range splits of a
// sample group added in split order, one thread per element (float4 where the row length allows it)
__global__ __launch_bounds__(256) void wgrad_det_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, long long mn,
                                                               int groups, int spg, int acc, int vec) {
    const long long total = (long long)groups * mn;
    if (vec) {
        const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i >= total) return;
        const long long g = i / mn, e = i - g * mn;
        float4 s = acc ? *reinterpret_cast<const float4*>(dw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* sp = slabs + (size_t)g * spg * mn + e;
        for (int sl = 0; sl < spg; ++sl) {
            const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)sl * mn);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dw + i) = s;
    } else {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= total) return;
        const long long g = i / mn, e = i - g * mn;
        float s = acc ? dw[i] : 0.f;
        const float* sp = slabs + (size_t)g * spg * mn + e;
        for (int sl = 0; sl < spg; ++sl) s += sp[(size_t)sl * mn];
        dw[i] = s;
    }
}
int launch_wgrad_det_reduce(const float* slabs, float* dw, long long mn, int groups, int spg, int accumulate, hipStream_t st) {
    const int vec = ((mn & 3) == 0 && ((((uintptr_t)slabs) | ((uintptr_t)dw)) & 15) == 0) ? 1 : 0;
    const long long total = (long long)groups * mn;
    const long long threads = vec ? total / 4 : total;
    hipLaunchKernelGGL(wgrad_det_reduce_kernel, dim3((unsigned)ceil_div(threads, 256LL)), dim3(256), 0, st, slabs, dw, mn, groups, spg,
                       accumulate ? 1 : 0, vec);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// ---------------------------------------------------------------- epilogue pass of split-K launches
__global__ __launch_bounds__(256) void act_mask_kernel(float* __restrict__ out, const float* __restrict__ mask, long long n, int relu,
                                                       int pack) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<float4*>(out + i);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (mask) {
            const float4 m = *reinterpret_cast<const float4*>(mask + i);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (pack) { v.x = pack_split(v.x); v.y = pack_split(v.y); v.z = pack_split(v.z); v.w = pack_split(v.w); }
        *reinterpret_cast<float4*>(out + i) = v;
    } else {
        for (long long k = i; k < n; ++k) {
            float v = out[k];
            if (relu) v = fmaxf(v, 0.f);
            if (mask) v = mask[k] > 0.f ? v : 0.f;
            out[k] = pack ? pack_split(v) : v;
        }
    }
}
// pack != 0: the finished values are rewritten as split words (ConvArgs::out_split of a split-K launch)
int launch_act_mask(float* out, const float* mask, long long n, int relu, hipStream_t st, int pack) {
    if (n <= 0 || (!relu && !mask && !pack)) return AG2V_OK;
    if ((((uintptr_t)out) & 15) || (mask && (((uintptr_t)mask) & 15))) return AG2V_EINVAL;
    hipLaunchKernelGGL(act_mask_kernel, dim3((unsigned)ceil_div(ceil_div(n, 4LL), 256LL)), dim3(256), 0, st, out, mask, n, relu, pack);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// ---------------------------------------------------------------- convolutions with <= 4 output rows (stride 1)
// The 3-channel data gradient of the decoder's first convolution (512 -> 3 at 256x256), conv_img / conv_flow (64 -> 3 / 2) and
// the discriminators' last layer (512 -> 1) have 1-4 GEMM rows: on a 32-row MFMA tile > 87 % of the matrix work is padding
// (6 TF measured).  Here a thread owns 4 consecutive output pixels x all rows and walks (c, r) with plain FMAs: S + 3 input
// values feed 4 S MR FMAs; weights (k-major [c][r][s][m]) are wave-uniform scalar loads; the input is read through the L1.
template <int MR, int S>   // MR = exact row count, S x S taps
__global__ __launch_bounds__(256) void conv_fewrow_kernel(ConvArgs p) {
    // lane = group of 4 output pixels, wave = quarter of the input channels; the four partial sums meet in LDS
    __shared__ float red[3][MR * 4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the weight reads become scalar loads
    const int groups_x = (p.OW + 3) >> 2;
    const long long total = (long long)p.N * p.OH * groups_x;
    // consecutive image rows share S-1 of their S input rows: keep them on one XCD's L2 (blocks are dealt round-robin)
    long long gid = (long long)xcd_remap(blockIdx.x, gridDim.x) * 64 + lane;
    const bool live = gid < total;
    if (!live) gid = total - 1;
    const int xg = (int)(gid % groups_x);
    const long long t = gid / groups_x;
    const int oy = (int)(t % p.OH), n = (int)(t / p.OH);
    const int ox0 = xg * 4;
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.b_src, p.b_bytes);
    float acc[MR][4];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    // byte offsets; an invalid column / row contributes 0xF0000000, so the sum stays >= 0xE0000000 > buffer size (launcher)
    const int x0 = ox0 - p.pad_w;
    unsigned coff[S + 3], roff[S];
#pragma unroll
    for (int i = 0; i < S + 3; ++i) coff[i] = ((unsigned)(x0 + i) < (unsigned)p.W) ? (unsigned)(x0 + i) * 4u : 0xF0000000u;
    const int HW = p.H * p.W;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        const int iy = oy - p.pad + r;
        roff[r] = ((unsigned)iy < (unsigned)p.H) ? (unsigned)(n * p.C * HW + iy * p.W) * 4u : 0xF0000000u;
    }
    const float* __restrict__ wt = p.a_src;
    const int cper = (p.C + 3) >> 2;
    const int c_lo = wave * cper, c_hi = min(p.C, c_lo + cper);
    for (int c = c_lo; c < c_hi; ++c) {
        const unsigned cb = (unsigned)(c * HW) * 4u;
        float v[S][S + 3];
#pragma unroll
        for (int r = 0; r < S; ++r)
#pragma unroll
            for (int i = 0; i < S + 3; ++i) v[r][i] = bload_f32(brs, roff[r] + cb + coff[i]);
        const float* wc = wt + (size_t)c * (S * S * MR);
#pragma unroll
        for (int r = 0; r < S; ++r)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const float wv = wc[(r * S + s2) * MR + m];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(wv, v[r][s2 + j], acc[m][j]);
                }
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave - 1][m * 4 + j][lane] = acc[m][j];
    }
    __syncthreads();
    if (wave > 0 || !live) return;
    const int plane = p.OH * p.OW;
    const float gs = p.gscale ? p.gscale[n / p.group_n] : 1.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float bv = p.bias ? p.bias[m] : 0.f;
        const size_t o = ((size_t)n * MR + m) * plane + (size_t)oy * p.OW + ox0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sum = acc[m][j] + (red[0][m * 4 + j][lane] + red[1][m * 4 + j][lane] + red[2][m * 4 + j][lane]);
            if (ox0 + j < p.OW) {
                float v = fmaf(sum, gs, bv) + (p.residual ? p.residual[o + j] : 0.f);
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask) v = (p.mask[o + j] > 0.f) ? v : 0.f;
                p.out[o + j] = v;
            }
        }
    }
}
// The same convolutions, second form (filter 3 x 3 pad 1, 4 x 4 pad 2, 1 x 1): the kernel above issues S (S + 3) dword loads per
// input channel for 4 S S MR FMAs and is bound by the CU's address unit (~10 cycles per dword wave-load: conv_img 13 TF, the
// discriminators' 512 -> 1 layer 4 TF).  Here a lane owns 4 pixels x TWO output rows and fetches ONE aligned 16-byte quad per input
// row (S + 1 rows); the pad_w columns left and S - 1 - pad_w columns right of it come from the neighbouring lanes' quads
// (ds_bpermute: the lanes of a wave walk consecutive quads of the (image, row pair, quad) order).  Lanes 0 and 63 of a wave only
// feed their neighbours (a wave covers 62 quads); the next channel's quads are in flight while the current one is multiplied.
template <int MR, int S, int PW>
__global__ __launch_bounds__(256) void conv_fewrow2_kernel(ConvArgs p) {
    constexpr int NR = S + 1, RW = S - 1 - PW;          // input rows per lane; columns taken from the right neighbour
    static_assert(PW <= 3 && RW >= 0 && RW <= 3, "window");
    __shared__ float red[3][MR * 8][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QG = max((p.OW + 3) >> 2, (p.W + 3) >> 2);
    const int RPN = (p.OH + 1) >> 1;
    const long long total = (long long)p.N * RPN * QG;
    const long long gq = (long long)xcd_remap(blockIdx.x, gridDim.x) * 62 + lane - 1;
    const bool inrange = gq >= 0 && gq < total;
    const long long g = inrange ? gq : 0;
    const int q = (int)(g % QG);
    const long long t = g / QG;
    const int rp = (int)(t % RPN), n = (int)(t / RPN);
    const int oy0 = 2 * rp, ox0 = 4 * q;
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.b_src, p.b_bytes);
    const int HW = p.H * p.W;
    unsigned roff[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int iy = oy0 - p.pad + r;
        roff[r] = (inrange && (unsigned)iy < (unsigned)p.H && ox0 < p.W) ? (unsigned)(n * p.C * HW + iy * p.W + ox0) * 4u : 0xF0000000u;
    }
    const bool has_left = q > 0, has_right = q < QG - 1;
    const bool wtail = (p.W & 3) != 0;                    // (uniform) the last quad of an input row is partial
    bool own_ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) own_ok[e] = ox0 + e < p.W;
    const int lsrc = ((lane + 63) & 63) * 4, rsrc = ((lane + 1) & 63) * 4;
    float acc[MR][2][4];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][o][j] = 0.f;
    const float* __restrict__ wt = p.a_src;
    const int cper = (p.C + 3) >> 2;
    const int c_lo = wave * cper, c_hi = min(p.C, c_lo + cper);
    u32x4 L[NR], Ln[NR];
    if (c_lo < c_hi) {
#pragma unroll
        for (int r = 0; r < NR; ++r) Ln[r] = bload_u32x4(brs, roff[r] + (unsigned)(c_lo * HW) * 4u);
    }
    for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
        for (int r = 0; r < NR; ++r) L[r] = Ln[r];
        if (c + 1 < c_hi) {
#pragma unroll
            for (int r = 0; r < NR; ++r) Ln[r] = bload_u32x4(brs, roff[r] + (unsigned)((c + 1) * HW) * 4u);
        }
        float v[NR][S + 3];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            unsigned own[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) own[e] = (wtail && !own_ok[e]) ? 0u : L[r][e];
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const unsigned x = (unsigned)__builtin_amdgcn_ds_bpermute(lsrc, (int)own[4 - PW + i]);
                v[r][i] = has_left ? __uint_as_float(x) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[r][PW + e] = __uint_as_float(own[e]);
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const unsigned x = (unsigned)__builtin_amdgcn_ds_bpermute(rsrc, (int)own[i]);
                v[r][PW + 4 + i] = has_right ? __uint_as_float(x) : 0.f;
            }
        }
        const float* wc = wt + (size_t)c * (S * S * MR);
#pragma unroll
        for (int r = 0; r < S; ++r)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const float wv = wc[(r * S + s2) * MR + m];
#pragma unroll
                    for (int o = 0; o < 2; ++o)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][o][j] = fmaf(wv, v[o + r][s2 + j], acc[m][o][j]);
                }
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wave - 1][(m * 2 + o) * 4 + j][lane] = acc[m][o][j];
    }
    __syncthreads();
    if (wave > 0 || !inrange || lane == 0 || lane == 63 || ox0 >= p.OW) return;
    // output placement (ConvArgs::o_mul ...): pixel (oy, ox) of the GEMM's pixel space is (oy o_mul + o_add_y, ox o_mul + o_add_x) of the
    // [o_H, o_W] output -- the parity classes of a strided data gradient write every o_mul-th pixel
    const int plane = p.o_H * p.o_W;
    const float gs = p.gscale ? p.gscale[n / p.group_n] : 1.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float bv = p.bias ? p.bias[m] : 0.f;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int yy = (oy0 + o) * p.o_mul + p.o_add_y;
            if (oy0 + o >= p.OH || yy >= p.o_H) continue;
            const size_t ob = ((size_t)n * MR + m) * plane + (size_t)yy * p.o_W;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ri = (m * 2 + o) * 4 + j;
                const float sum = acc[m][o][j] + (red[0][ri][lane] + red[1][ri][lane] + red[2][ri][lane]);
                const int xx = (ox0 + j) * p.o_mul + p.o_add_x;
                if (ox0 + j < p.OW && xx < p.o_W) {
                    float y = fmaf(sum, gs, bv) + (p.residual ? p.residual[ob + xx] : 0.f);
                    if (p.relu) y = fmaxf(y, 0.f);
                    if (p.mask) y = (p.mask[ob + xx] > 0.f) ? y : 0.f;
                    p.out[ob + xx] = y;
                }
            }
        }
    }
}
static bool g_fewrow2_last = false;      // which of the two forms the last few-row launch took (kernel name for the timing tables)
template <int MR>
static bool launch_fewrow2(const ConvArgs& a, hipStream_t st) {
    static const bool on = true;
    g_fewrow2_last = false;
    if (!on || a.R != a.S || a.pad != a.pad_w) return false;
    g_fewrow2_last = (a.S == 3 && a.pad_w == 1) || (a.S == 4 && (a.pad_w == 2 || a.pad_w == 1)) || (a.S == 1 && a.pad_w == 0) ||
                     (a.S == 2 && (a.pad_w == 0 || a.pad_w == 1));
    const int QG = ((a.OW + 3) >> 2) > ((a.W + 3) >> 2) ? ((a.OW + 3) >> 2) : ((a.W + 3) >> 2);
    const long long total = (long long)a.N * ((a.OH + 1) >> 1) * QG;
    const dim3 grid((unsigned)ceil_div(total, 62LL));
    if (a.S == 3 && a.pad_w == 1) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 3, 1>), grid, dim3(256), 0, st, a); return true; }
    if (a.S == 4 && a.pad_w == 2) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 4, 2>), grid, dim3(256), 0, st, a); return true; }
    if (a.S == 4 && a.pad_w == 1) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 4, 1>), grid, dim3(256), 0, st, a); return true; }
    if (a.S == 1 && a.pad_w == 0) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 1, 0>), grid, dim3(256), 0, st, a); return true; }
    // 2 x 2: the parity classes of a 4 x 4 stride-2 data gradient (the discriminators' first layer, 64 -> 3 channels)
    if (a.S == 2 && a.pad_w == 0) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 2, 0>), grid, dim3(256), 0, st, a); return true; }
    if (a.S == 2 && a.pad_w == 1) { hipLaunchKernelGGL((conv_fewrow2_kernel<MR, 2, 1>), grid, dim3(256), 0, st, a); return true; }
    return false;
}
template <int MR>
static bool launch_fewrow(const ConvArgs& a, hipStream_t st, bool mapped) {
    if (a.R != a.S) return false;
    if (launch_fewrow2<MR>(a, st)) return true;
    if (mapped) return false;            // only the second form places its output (ConvArgs::o_mul ...)
    const long long total = (long long)a.N * a.OH * ((a.OW + 3) >> 2);
    const dim3 grid((unsigned)ceil_div(total, 64LL));
    switch (a.S) {
        case 1: hipLaunchKernelGGL((conv_fewrow_kernel<MR, 1>), grid, dim3(256), 0, st, a); return true;
        case 3: hipLaunchKernelGGL((conv_fewrow_kernel<MR, 3>), grid, dim3(256), 0, st, a); return true;
        case 4: hipLaunchKernelGGL((conv_fewrow_kernel<MR, 4>), grid, dim3(256), 0, st, a); return true;
        default: return false;
    }
}

static int launch_igemm(ConvArgs& a, hipStream_t st, bool out_zeroed, bool split) {
    if (a.M <= 0 || a.NG <= 0 || a.KG <= 0 || a.R * a.S > 16) return AG2V_EINVAL;
    if (a.o_mul == 0) { a.o_mul = 1; a.o_add_y = 0; a.o_add_x = 0; a.o_H = a.OH; a.o_W = a.OW; }
    const long long dense_nstride = (long long)a.M * a.o_H * a.o_W;
    if (a.out_nstride == 0) a.out_nstride = dense_nstride;
    const bool dense_out = a.out_nstride == dense_nstride;
    if (split) return launch_igemm_bf16x3(a, st, out_zeroed);
    const bool mapped = !(a.o_mul == 1 && a.o_add_y == 0 && a.o_add_x == 0 && a.o_W == a.OW && a.o_H == a.OH);
    static const bool fewrow_mapped = true;
    if (dense_out && a.M <= 4 && a.stride == 1 && (!mapped || (fewrow_mapped && a.S == 2 && a.R == 2 && a.pad == a.pad_w)) &&
        (long long)a.C * a.H * a.W * 4 < 0x10000000LL && true) {
        // The kernel flags an out-of-image tap by adding 0xF0000000 to its byte offset, which only stays out of range while the
        // valid part of the offset is below 0x10000000: at most 256 MB of input per launch (a frame-batched [120, 64, 256, 256]
        // input is 2 GB: beyond 256 MB the padding taps of the few-row kernel read real data) -> sample chunks.
        const long long per = (long long)a.C * a.H * a.W * 4;
        int nmax = (int)(0x0FFFFFFFLL / per);
        if (a.gscale) nmax -= nmax % a.group_n;
        if (nmax >= 1) {
            const size_t os = (size_t)a.M * a.o_H * a.o_W;
            bool ok = true;
            for (int n0 = 0; n0 < a.N && ok; n0 += nmax) {
                ConvArgs c = a;
                c.N = (a.N - n0 < nmax) ? a.N - n0 : nmax;
                c.b_src = a.b_src + (size_t)n0 * a.C * a.H * a.W;
                c.out = a.out + (size_t)n0 * os;
                c.residual = a.residual ? a.residual + (size_t)n0 * os : nullptr;
                c.mask = a.mask ? a.mask + (size_t)n0 * os : nullptr;
                c.gscale = a.gscale ? a.gscale + n0 / a.group_n : nullptr;
                c.NG = c.N * a.OH * a.OW;
                c.b_bytes = (unsigned)(per * c.N);
                ok = a.M == 1 ? launch_fewrow<1>(c, st, mapped) : a.M == 2 ? launch_fewrow<2>(c, st, mapped)
                   : a.M == 3 ? launch_fewrow<3>(c, st, mapped) : launch_fewrow<4>(c, st, mapped);
                if (!ok && n0 > 0) return AG2V_EINVAL;      // (cannot happen: the filter shape decides and is the same for all chunks)
            }
            if (ok) { set_last_conv_kernel(g_fewrow2_last ? "conv_fewrow2_kernel" : "conv_fewrow_kernel"); AG2V_CHECK_LAUNCH(); return AG2V_OK; }
        }
    }
    const int bm = (a.M <= 32) ? 32 : (a.M <= 64 ? 64 : 128);
    const int bk = 16;
    const int bn = (bm == 128) ? 128 : 256;
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = ceil_div(a.NG, bn);
    // Layers with few output pixels (8x8 ... 32x32 maps) have fewer tiles than the chip has CUs while their K loop is
    // thousands of steps long: split K over grid.y and accumulate with fp32 atomics into the zeroed output.
    const int tiles = a.tiles_m * a.tiles_n, nk = ceil_div(a.KG, bk);
    a.splitk = 1; a.kchunk = nk;
    if (tiles < 384 && nk >= 16 && !a.det && dense_out) {      // (deterministic: no split-K, no atomics)
        int sk = ceil_div(768, tiles);
        if (sk > nk / 8) sk = nk / 8;
        if (sk > 1) {
            a.kchunk = ceil_div(nk, sk);
            a.splitk = ceil_div(nk, a.kchunk);
        }
    }
    if (a.splitk > 1 && !out_zeroed) {
        if (hipMemsetAsync(a.out, 0, (size_t)a.N * a.M * a.o_H * a.o_W * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    if (a.splitk > 1 && a.residual == a.out) a.residual = nullptr;   // accumulating launch: the atomics already add onto out
    dim3 grid(tiles, a.splitk);
    if (bm == 32) {
        set_last_conv_kernel("conv_igemm_kernel<32, 256, 1, 4>");
        hipLaunchKernelGGL((conv_igemm_kernel<32, 256, 1, 4>), grid, dim3(256), 0, st, a);
    } else if (bm == 64) {
        set_last_conv_kernel("conv_igemm_kernel<64, 256, 1, 4>");
        hipLaunchKernelGGL((conv_igemm_kernel<64, 256, 1, 4>), grid, dim3(256), 0, st, a);
    } else {
        set_last_conv_kernel("conv_igemm_kernel<128, 128, 2, 2>");
        hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
    }
    AG2V_CHECK_LAUNCH();
    if (a.splitk > 1 && (a.relu || a.mask))
        return launch_act_mask(a.out, a.mask, (long long)a.N * a.M * a.o_H * a.o_W, a.relu, st);
    return AG2V_OK;
}

// ---------------------------------------------------------------- data gradient of a convolution with <= 4 input channels
// dx[n,c,y,x] = sum_{r,s} t[n, (c R + r) S + s, y + pad - r, x + pad - s],  t = W'^T dy as a 1x1 convolution (W' = w viewed as
// [K][C R S]): the K-deep reduction becomes a plain GEMM over contiguous pixels (no gather, dy read once with 16-byte loads) and
// the filter footprint a sum of C R S shifted planes.  8 x 512 x 256 x 256 -> 3 channels: 1.02 ms (few-row FMA kernel) -> 0.45 ms.
__global__ __launch_bounds__(256) void col2im_few_kernel(const float* __restrict__ t, float* __restrict__ dx, int N, int C, int H, int W,
                                                         int OH, int OW, int R, int S, int pad, int acc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)N * C * H * W;
    if (i >= total) return;
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int c = (int)((i / ((long long)W * H)) % C);
    const int n = (int)(i / ((long long)W * H * C));
    const float* tp = t + ((size_t)n * C * R * S + (size_t)c * R * S) * OH * OW;
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
        const int oy = y + pad - r;
        if ((unsigned)oy >= (unsigned)OH) continue;
        for (int q = 0; q < S; ++q) {
            const int ox = x + pad - q;
            if ((unsigned)ox < (unsigned)OW) s += tp[(size_t)(r * S + q) * OH * OW + (size_t)oy * OW + ox];
        }
    }
    dx[i] = acc ? dx[i] + s : s;
}

extern "C" size_t ag2v_conv2d_workspace_floats(int C, int K, int R, int S) { return (size_t)C * K * R * S; }

// Samples per launch so that the gathered operand of a bf16x3 launch stays below the 4 GiB a raw buffer descriptor covers (the
// frame-batched generator convolves [T*B, C, H, W] tensors of up to 16 GB); a multiple of group_n.  0: one sample is too large.
// (AG2V_CHUNK_BYTES lowers the limit so that tests can exercise the chunk loop at small sizes.)
static long long chunk_limit() {
    const char* e = getenv("AG2V_CHUNK_BYTES");
    const long long v = e ? atoll(e) : 0;
    return (v > 0 && v < kBufMax) ? v : kBufMax;
}
static int chunk_samples(long long bytes_per_sample, long long bytes_per_sample2, int group_n) {
    const long long per = bytes_per_sample > bytes_per_sample2 ? bytes_per_sample : bytes_per_sample2;
    long long n = chunk_limit() / (per > 0 ? per : 1);
    if (n > (1 << 30)) n = 1 << 30;
    n -= n % group_n;
    return (int)n;
}

// y = conv(x, w) * gscale[n / group_n] (+bias) (+residual).  workspace: ag2v_conv2d_workspace_floats() floats (k-major copy of the
// weights).  relu != 0: y = max(., 0) (the ReLU behind SPADE's mlp_shared convolution, normalization.py:92-95)
extern "C" int ag2v_conv2d_fwd_ex(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                  int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, float* workspace,
                                  void* stream, ag2v_conv_opts* o) {
    if (!x || !w || !y || !workspace || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || R * S > 16) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const ConvCfg cfg = conv_cfg(o);
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
    if ((long long)C * R * S >= (1LL << 31) || (long long)C * H * W >= (1LL << 31) || (long long)K * OH * OW >= (1LL << 31))
        return AG2V_EINVAL;
    const int KG = C * R * S;
    const long long abytes = (long long)KG * K * 4, xs = (long long)C * H * W, ys = (long long)K * OH * OW;
    // (a reduction shorter than two K-tiles is all epilogue: the exact kernels' smaller tiles keep every wave busy there --
    // D's first convolution, 3 image channels x 4x4 -> 64 rows with the layout term as residual)
    static const bool short_exact = true;
    int nmax = N;
    bool split = use_bf16x3(cfg, K) && abytes < kBufMax && !(short_exact && KG < 64 && K < 128);
    // precision 2: fp16 3-term split for THIS forward (operands of O(1) magnitude: the flow network); same tiles as bf16x3
    const bool h16 = cfg.precision == 2 && split;
    if (h16 && (cfg.in_split || cfg.out_split)) return AG2V_EINVAL;
    if (split && xs * 4 * N >= chunk_limit()) {
        nmax = chunk_samples(xs * 4, 0, cfg.gscale ? cfg.group_n : 1);
        if (nmax <= 0) { split = false; nmax = N; }
    }
    if ((cfg.in_split || cfg.out_split) && !split) return AG2V_EINVAL;   // split words: bf16x3 kernels only (ag2v_conv2d_split_input_ok)
    // exact arithmetic with >= 128 rows: the wave-specialised pipeline on fp32 operands (same chunking / descriptors as `split`)
    static const bool f32ws_on = true;
    bool f32ws = !split && f32ws_on && K >= 128 && KG >= 64 && abytes < kBufMax;
    if (f32ws && xs * 4 * N >= chunk_limit()) {
        nmax = chunk_samples(xs * 4, 0, cfg.gscale ? cfg.group_n : 1);
        if (nmax <= 0) { f32ws = false; nmax = N; }
    }
    if (cfg.out_split && residual) return AG2V_EINVAL;
    if ((long long)nmax * OH * OW >= (1LL << 31)) {         // the GEMM's pixel index is a 32-bit int
        nmax = (int)(((1LL << 31) - 1) / ((long long)OH * OW));
        if (cfg.gscale) nmax -= nmax % cfg.group_n;
        if (nmax <= 0) return AG2V_EINVAL;
    }
    const bool tapmaj = igemm_tap_major(C, split);
    if (tapmaj)
        hipLaunchKernelGGL(pack_fwd_tap_kernel, dim3(C / 32, ceil_div(K, 8)), dim3(256), 0, st, w, workspace, K, C, R, S, h16 ? 1 : 0);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(KG, 32), ceil_div(K, 32)), dim3(256), 0, st, w, workspace, K, KG,
                           h16 ? 2 : (split ? 1 : 0));
    // ag2v_conv_opts::out_mstride: every 128 output channels are their own dense [N, 128, OH, OW] tensor, out_mstride floats apart
    // (several convolutions of ONE input as one launch).  Only the LDS-DMA kernels place their output so: refused before anything is
    // launched when this call would not run on them (ag2v_conv2d_fwd_mtile_ok answers the same question up front)
    const long long ms = cfg.out_mstride;
    const long long ys_tile = 128LL * OH * OW;
    if (ms < 0 || (ms > 0 && (residual || !split || !igemm_tap_major(C, split) || K % 128))) return AG2V_EINVAL;
    if (ms > 0) {
        for (int n0 = 0; n0 < N; n0 += nmax) {
            ConvArgs t{};
            t.N = (N - n0 < nmax) ? N - n0 : nmax; t.C = C; t.H = H; t.W = W; t.K = K; t.R = R; t.S = S; t.stride = stride; t.pad = pad; t.pad_w = pad;
            t.OH = OH; t.OW = OW; t.M = K; t.NG = t.N * OH * OW; t.KG = KG; t.b_bytes = (unsigned)(xs * 4 * t.N); t.tap_major = 1;
            t.b_split = cfg.in_split; t.out_split = cfg.out_split; t.det = cfg.det; t.h16 = h16 ? 1 : 0; t.o_mul = 1; t.o_H = OH; t.o_W = OW;
            t.out_mstride = ms;
            if (!igemm_halo_ok(t)) return AG2V_EINVAL;
        }
    }
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int nc = (N - n0 < nmax) ? N - n0 : nmax;
        ConvArgs a{};
        a.b_src = x + (size_t)n0 * xs; a.out = y + (size_t)n0 * (ms > 0 ? ys_tile : ys); a.bias = bias;
        if (ms > 0) { a.out_mstride = ms; a.out_nstride = ys_tile; }
        a.residual = residual ? residual + (size_t)n0 * ys : nullptr; a.relu = relu ? 1 : 0;
        a.gscale = cfg.gscale ? cfg.gscale + n0 / cfg.group_n : nullptr; a.group_n = cfg.group_n;
        a.N = nc; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.pad_w = pad;
        a.OH = OH; a.OW = OW;
        a.M = K; a.NG = nc * OH * OW; a.KG = KG;
        a.a_bytes = (unsigned)abytes; a.b_bytes = (split || f32ws) ? (unsigned)(xs * 4 * nc) : 0u;
        a.tap_major = tapmaj ? 1 : 0;
        a.a_src = workspace;
        a.b_split = cfg.in_split; a.out_split = cfg.out_split;
        a.f32 = f32ws ? 1 : 0;
        a.det = cfg.det;
        a.h16 = h16 ? 1 : 0;
        a.oscale = h16 ? 1.0f / AG2V_H16_WSCALE : 0.f;
        int rc;
        if (ms > 0) {
            a.o_mul = 1; a.o_H = OH; a.o_W = OW;
            rc = launch_igemm_halo(a, st);
        } else {
            rc = launch_igemm(a, st, false, split || f32ws);
        }
        if (rc != AG2V_OK) return rc;
    }
    if (o) o->kernel = g_last_conv_kernel;
    return AG2V_OK;
}
// 1 when ag2v_conv2d_fwd_ex accepts ag2v_conv_opts::out_mstride for this call (3x3 stride-1 convolution with K % 128 == 0 that runs on the
// LDS-DMA kernels' 128-row tile in every sample chunk), else 0
extern "C" int ag2v_conv2d_fwd_mtile_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows, int precision,
                                        int in_split, int out_split, int deterministic) {
    if (N <= 0 || C <= 0 || K <= 0 || K % 128 || stride <= 0 || R * S > 16) return 0;
    ag2v_conv_opts o{};
    o.precision = precision; o.min_rows = min_rows; o.in_split = in_split; o.out_split = out_split; o.deterministic = deterministic;
    const ConvCfg cfg = conv_cfg(&o);
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return 0;
    const int KG = C * R * S;
    const long long abytes = (long long)KG * K * 4, xs = (long long)C * H * W;
    if (xs >= (1LL << 31) || (long long)K * OH * OW >= (1LL << 31)) return 0;
    if (!(use_bf16x3(cfg, K) && abytes < kBufMax && !(KG < 64 && K < 128)) || !igemm_tap_major(C, true)) return 0;
    const bool h16 = cfg.precision == 2;
    if (h16 && (in_split || out_split)) return 0;
    int nmax = N;
    if (xs * 4 * N >= chunk_limit()) {
        nmax = chunk_samples(xs * 4, 0, 1);
        if (nmax <= 0) return 0;
    }
    if ((long long)nmax * OH * OW >= (1LL << 31)) nmax = (int)(((1LL << 31) - 1) / ((long long)OH * OW));
    if (nmax <= 0) return 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        ConvArgs t{};
        t.N = (N - n0 < nmax) ? N - n0 : nmax; t.C = C; t.H = H; t.W = W; t.K = K; t.R = R; t.S = S; t.stride = stride; t.pad = pad; t.pad_w = pad;
        t.OH = OH; t.OW = OW; t.M = K; t.NG = t.N * OH * OW; t.KG = KG; t.b_bytes = (unsigned)(xs * 4 * t.N); t.tap_major = 1;
        t.b_split = in_split ? 1 : 0; t.out_split = out_split ? 1 : 0; t.det = deterministic ? 1 : 0; t.h16 = h16 ? 1 : 0;
        t.o_mul = 1; t.o_H = OH; t.o_W = OW; t.out_mstride = 1;
        if (!igemm_halo_ok(t)) return 0;
    }
    return 1;
}
extern "C" int ag2v_conv2d_fwd_act(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                   int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, float* workspace,
                                   void* stream) {
    return ag2v_conv2d_fwd_ex(x, w, bias, residual, y, N, C, H, W, K, R, S, stride, pad, relu, workspace, stream, nullptr);
}
extern "C" int ag2v_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                               int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace,
                               void* stream) {
    return ag2v_conv2d_fwd_ex(x, w, bias, residual, y, N, C, H, W, K, R, S, stride, pad, 0, workspace, stream, nullptr);
}

// ---------------------------------------------------------------- stride-2 data gradient as ONE convolution + depth-to-space
// The four pixel classes (py, px) of a stride-2 data gradient are stride-1 convolutions of dy with <= 2 x 2 sub-filters.  Run one by
// one each is a GEMM with C rows (64 of a 128-row tile for the discriminators' 128 -> 64 layer) that scatters 4-byte stores to every
// second pixel of dx: 93-240 TF, 11.5 ms per c3 step.  When all four classes read the SAME 2 x 2 window of dy (4 x 4 pad 2, 3 x 3
// pad 1: discriminator.py:368-379, flows_generator.py:45-47) they are one stride-1 2 x 2 convolution to 4 C output channels
// (class-major rows, zero taps where a class's sub-filter is smaller) on the half-resolution grid -- full 256-row tiles, dense
// 16-byte stores into a workspace -- followed by a depth-to-space pass (mask, accumulate) over dx.
struct S2Geom { int Q, ay[2], rc[2]; bool ok; };      // per axis: grid size, first dx row of class p, sub-filter length of class p
static S2Geom s2_geom(int L, int R, int pad) {      // L: dx length along the axis
    S2Geom g{};
    g.ok = R <= 4;
    for (int p = 0; p < 2; ++p) {
        const int rc = (R - p + 1) / 2;
        const int q0 = (pad > p) ? (pad - p + 1) / 2 : 0;
        g.rc[p] = rc;
        g.ay[p] = q0 * 2 + p - pad;
        if (rc < 1 || rc > 2 || rc - 1 - q0 != 0 || g.ay[p] < 0 || g.ay[p] > 1 || g.ay[p] >= L) g.ok = false;
        const int n = g.ok ? (L - g.ay[p] + 1) / 2 : 0;
        if (n > g.Q) g.Q = n;
    }
    if (g.ay[0] == g.ay[1]) g.ok = false;
    return g;
}
// plane[((tap_major_row(k, i, i2; 2, 2) >> 3) * 4C + cls * C + c) * 8 + (k & 7)] = hi | lo of the class's tap at window position
// (i, i2), or 0 -- grid (ceil(C / 64), K / 8), as pack_dgrad_tap_kernel
__global__ __launch_bounds__(256) void pack_dgrad_s2_tap_kernel(const float* __restrict__ w, float* __restrict__ at, int K, int C, int R, int S,
                                                                int rc0, int rc1, int sc0, int sc1) {
    __shared__ float t[8][64 * 17];
    const int RS = R * S, RSP = RS | 1;
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 8;
    const int nc = min(64, C - c0);
    for (int j = 0; j < 8; ++j) {
        const float* src = w + ((size_t)(k0 + j) * C + c0) * RS;
        for (int i = threadIdx.x; i < nc * RS; i += 256) {
            const int c = i / RS;
            t[j][c * RSP + (i - c * RS)] = src[i];
        }
    }
    __syncthreads();
    const int M = 4 * C;
    uint4* hi_plane = reinterpret_cast<uint4*>(at);
    uint4* lo_plane = hi_plane + (size_t)(K * 4 / 8) * M;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int u = grp; u < 16; u += 4) {           // (class, window position)
        const int cls = u >> 2, i = (u >> 1) & 1, i2 = u & 1;
        const int py = cls >> 1, px = cls & 1;
        const int Rc = py ? rc1 : rc0, Sc = px ? sc1 : sc0;
        if (lane < nc) {
            float x[8];
            const bool live = i < Rc && i2 < Sc;
            const int q = lane * RSP + (py + 2 * (Rc - 1 - i)) * S + (px + 2 * (Sc - 1 - i2));
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = live ? t[e][q] : 0.f;
            uint4 hi, lo;
            split_planes8(x, hi, lo);
            const size_t idx = (size_t)(tap_major_row(k0, i, i2, 2, 2) >> 3) * M + (size_t)cls * C + (c0 + lane);
            hi_plane[idx] = hi;
            lo_plane[idx] = lo;
        }
    }
}
// dx[n, c, y, x] (+)= t[n, cls * C + c, (y - ay[py]) / 2, (x - ax[px]) / 2] for the class whose parity (y, x) has; mask as the epilogues
// grid (N * C, ceil(H * ceil(W / 2) / 512)), 256 threads: a thread writes two pixel pairs (2 t, 2 t + 1) of the plane, 256 pairs apart -- one
// value of each x class per pair, both read along consecutive qx.  32-bit index arithmetic (the first form spent its time in 64-bit
// div / mod chains: 4.5 ms per c3 step for 3 GB of traffic); the pairs of a plane are flattened so that odd widths (129, 65, 33) fill the waves.
__global__ __launch_bounds__(256) void depth_to_space_s2_kernel(const float* __restrict__ t, const float* __restrict__ mask, float* __restrict__ dx,
                                                                int C, int H, int W, int Qy, int Qx, int ay0, int ax0, int acc) {
    const unsigned nc = blockIdx.x;
    const unsigned n = nc / (unsigned)C, c = nc - n * (unsigned)C;
    const unsigned W2 = (unsigned)(W + 1) >> 1, P = (unsigned)H * W2;
    const size_t plane = (size_t)Qy * Qx;
    // x = 2 t + e has the class of its parity: px = (e - ax0) & 1, and qx = t for both (the class's first column is e)
    const int px_e0 = (0 - ax0) & 1, px_e1 = (1 - ax0) & 1;
    const float* tn = t + (size_t)n * 4 * C * plane + (size_t)c * plane;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned p = blockIdx.y * 512u + u * 256u + threadIdx.x;
        if (p >= P) return;
        const unsigned y = p / W2, tt = p - y * W2;
        const int py = ((int)y - ay0) & 1;                        // class 0 owns the rows of ay0's parity
        const int qy = ((int)y - (py ? 1 - ay0 : ay0)) >> 1;
        const float* r0 = tn + (size_t)(py * 2 + px_e0) * C * plane + (size_t)qy * Qx;
        const float* r1 = tn + (size_t)(py * 2 + px_e1) * C * plane + (size_t)qy * Qx;
        const size_t row = ((size_t)nc * H + y) * W;
        const unsigned x0 = 2 * tt;
        const bool two = x0 + 1 < (unsigned)W;
        float v0 = r0[tt];
        float v1 = two ? r1[tt] : 0.f;
        if (mask) {
            v0 = (mask[row + x0] > 0.f) ? v0 : 0.f;
            if (two) v1 = (mask[row + x0 + 1] > 0.f) ? v1 : 0.f;
        }
        if (acc) {
            v0 += dx[row + x0];
            if (two) v1 += dx[row + x0 + 1];
        }
        dx[row + x0] = v0;
        if (two) dx[row + x0 + 1] = v1;
    }
}
static bool s2_unified_ok(const ConvCfg& cfg, int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int OH, int OW) {
    static const bool on = true;
    if (!on || stride != 2 || R != S || cfg.out_nstride > 0 || cfg.out_split) return false;
    if (!use_bf16x3(cfg, C) || !igemm_tap_major(K, true) || C < 32) return false;
    const S2Geom gy = s2_geom(H, R, pad), gx = s2_geom(W, S, pad);
    if (!gy.ok || !gx.ok) return false;
    if ((long long)K * 4 * 4 * C * 4 >= kBufMax || (long long)4 * C * gy.Q * gx.Q >= (1LL << 31)) return false;
    (void)N; (void)OH; (void)OW;
    return true;
}

// dx = d conv / d x (* gscale[n / group_n]).  For each pixel class (py,px) of the stride one forward-kernel launch over dy with the
// class's flipped sub-filter.  workspace: ag2v_conv2d_dgrad_workspace_floats() floats.
// mask (nullable, same shape as dx): dx = (mask > 0) ? dx : 0 -- the backward of a ReLU that produced this convolution's input
// (SPADE: mlp_gamma / mlp_beta read relu(mlp_shared(seg)); mask = that activation), fused into the epilogue.
// accumulate != 0: dx += (data gradient) instead of dx = : several convolutions that read the same tensor (the three SPADE
// mlp_shared convolutions of a residual block) sum their input gradients without a separate pass per pair.  Not with a mask.
static bool few_channel_dgrad(int N, int C, int K, int R, int S, int OH, int OW, int stride, bool mask) {
    return C <= 4 && stride == 1 && !mask && K >= 64 && C * R * S <= 64 && true;
}
extern "C" size_t ag2v_conv2d_dgrad_workspace_floats(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    size_t n = (size_t)C * K * R * S;
    // few input channels: the [N, C R S, OH, OW] intermediate of the 1x1-GEMM + shifted-plane-sum form lives in the workspace too
    if (OH > 0 && OW > 0 && few_channel_dgrad(N, C, K, R, S, OH, OW, stride, false)) n += (size_t)N * C * R * S * OH * OW;
    if (OH > 0 && OW > 0 && stride == 2 && R == S && R <= 4) {      // the one-convolution form of a stride-2 data gradient (if the call takes it)
        const S2Geom gy = s2_geom(H, R, pad), gx = s2_geom(W, S, pad);
        if (gy.ok && gx.ok) {
            const size_t need = (size_t)16 * K * C + (size_t)N * 4 * C * gy.Q * gx.Q;
            if (need > n) n = need;
        }
    }
    return n;
}
extern "C" int ag2v_conv2d_dgrad_ex(const float* dy, const float* w, const float* mask, float* dx,
                                    int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int accumulate,
                                    float* workspace, size_t workspace_floats, void* stream, ag2v_conv_opts* o) {
    if (!dy || !w || !dx || !workspace || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || R * S > 16) return AG2V_EINVAL;
    if (accumulate && mask) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const ConvCfg cfg = conv_cfg(o);
    note_chan_sums_done(0);
    if (o) o->chan_sums_done = 0;
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
    if ((long long)K * OH * OW >= (1LL << 31) || (long long)C * H * W >= (1LL << 31)) return AG2V_EINVAL;
    if (workspace_floats < (size_t)C * K * R * S) return AG2V_EINVAL;
    const long long dys = (long long)K * OH * OW, dxs = (long long)C * H * W;
    const size_t t_floats = (size_t)N * C * R * S * OH * OW;
    if ((cfg.dy_split || cfg.out_split) && (few_channel_dgrad(N, C, K, R, S, OH, OW, stride, mask != nullptr) || !use_bf16x3(cfg, C)))
        return AG2V_EINVAL;                      // split words are only read / written by the bf16x3 kernels (ag2v_conv2d_split_dy_ok)
    if (cfg.out_split && (accumulate || stride != 1)) return AG2V_EINVAL;   // dx as split words: one class, written once
    // dx as a channel slice of a wider buffer (ag2v_conv_opts::out_nstride): one class, written once, by the general kernels
    const long long dxo = cfg.out_nstride > 0 ? cfg.out_nstride : dxs;
    if (dxo != dxs && (dxo < dxs || accumulate || stride != 1 || few_channel_dgrad(N, C, K, R, S, OH, OW, stride, mask != nullptr)))
        return AG2V_EINVAL;
    if (few_channel_dgrad(N, C, K, R, S, OH, OW, stride, mask != nullptr) && (long long)N * OH * OW < (1LL << 31) &&
        workspace_floats >= (size_t)C * K * R * S + t_floats) {
        // few input channels: dx[n,c,y,x] = sum_{r,s} t[n, (c R + r) S + s, y + pad - r, x + pad - s], t = W'^T dy as a 1x1
        // convolution over contiguous pixels + a sum of C R S shifted planes (col2im_few_kernel); t is caller workspace
        const int J = C * R * S;
        float* t = workspace + (size_t)C * K * R * S;
        ConvArgs f{};
        f.a_src = w;                      // w[K][C R S] is the k-major operand of the 1x1 convolution as it stands
        f.b_src = dy; f.out = t;
        f.N = N; f.C = K; f.H = OH; f.W = OW; f.K = J; f.R = 1; f.S = 1; f.stride = 1; f.pad = 0; f.pad_w = 0; f.OH = OH; f.OW = OW;
        f.M = J; f.NG = N * OH * OW; f.KG = K;
        f.a_bytes = (unsigned)((long long)K * J * 4); f.b_bytes = 0;
        f.gscale = cfg.gscale; f.group_n = cfg.group_n;
        f.det = cfg.det;
        int rc = launch_igemm(f, st, false, false);
        if (rc == AG2V_OK) {
            const long long total = (long long)N * C * H * W;
            hipLaunchKernelGGL(col2im_few_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, t, dx, N, C, H, W, OH, OW, R, S,
                               pad, accumulate ? 1 : 0);
            if (hipGetLastError() != hipSuccess) rc = AG2V_EHIP;
        }
        if (o) o->kernel = g_last_conv_kernel;
        return rc;
    }
    if (s2_unified_ok(cfg, N, C, H, W, K, R, S, stride, pad, OH, OW)) {
        const S2Geom gy = s2_geom(H, R, pad), gx = s2_geom(W, S, pad);
        const size_t wfl = (size_t)16 * K * C, per_t = (size_t)4 * C * gy.Q * gx.Q;
        int nmax = N;
        if (dys * 4 * N >= chunk_limit() || (long long)per_t * 4 * N >= chunk_limit())
            nmax = chunk_samples(dys * 4, (long long)per_t * 4, cfg.gscale ? cfg.group_n : 1);
        if ((long long)nmax * gy.Q * gx.Q >= (1LL << 31)) nmax = (int)(((1LL << 31) - 1) / ((long long)gy.Q * gx.Q));
        if (cfg.gscale) nmax -= nmax % cfg.group_n;
        if (nmax > 0 && workspace_floats >= wfl + per_t * (size_t)(nmax < N ? nmax : N)) {
            float* at = workspace;
            float* tmp = workspace + wfl;
            hipLaunchKernelGGL(pack_dgrad_s2_tap_kernel, dim3(ceil_div(C, 64), K / 8), dim3(256), 0, st, w, at, K, C, R, S, gy.rc[0], gy.rc[1],
                               gx.rc[0], gx.rc[1]);
            for (int n0 = 0; n0 < N; n0 += nmax) {
                const int nc = (N - n0 < nmax) ? N - n0 : nmax;
                ConvArgs f{};
                f.tap_major = 1;
                f.a_src = at; f.b_src = dy + (size_t)n0 * dys; f.out = tmp;
                f.N = nc; f.C = K; f.H = OH; f.W = OW; f.K = 4 * C; f.R = 2; f.S = 2; f.stride = 1; f.pad = 0; f.pad_w = 0;
                f.OH = gy.Q; f.OW = gx.Q;
                f.M = 4 * C; f.NG = nc * gy.Q * gx.Q; f.KG = K * 4;
                f.a_bytes = (unsigned)((long long)K * 4 * 4 * C * 4); f.b_bytes = (unsigned)(dys * 4 * nc);
                f.gscale = cfg.gscale ? cfg.gscale + n0 / cfg.group_n : nullptr; f.group_n = cfg.group_n;
                f.b_split = cfg.dy_split;
                f.det = cfg.det;
                const int rc = launch_igemm(f, st, false, true);
                if (rc != AG2V_OK) return rc;
                hipLaunchKernelGGL(depth_to_space_s2_kernel, dim3((unsigned)(nc * C), (unsigned)ceil_div(H * ((W + 1) / 2), 512)), dim3(256), 0, st, tmp,
                                   mask ? mask + (size_t)n0 * dxs : nullptr, dx + (size_t)n0 * dxs, C, H, W, gy.Q, gx.Q, gy.ay[0], gx.ay[0],
                                   accumulate ? 1 : 0);
                if (hipGetLastError() != hipSuccess) return AG2V_EHIP;
            }
            if (o) o->kernel = g_last_conv_kernel;
            return AG2V_OK;
        }
    }
    // With stride > 1 several classes write disjoint pixels of dx (one of them may be empty when R < stride) and a class
    // may take the atomic split-K epilogue: zero dx once up front.  Stride 1 is a single class covering every pixel.
    const bool multi = (stride > 1);
    if (multi && !accumulate && hipMemsetAsync(dx, 0, (size_t)N * C * H * W * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    float* at = workspace;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            // class (py,px): pixels y = ay + stride*q, x = ax + stride*q' with (y+pad)%stride == py; they see the taps
            // r = py + stride*r', r' < Rc, at dy row (y+pad-py)/stride - r' = q + q0y - r'.
            const int Rc = (R - py + stride - 1) / stride, Sc = (S - px + stride - 1) / stride;
            if (Rc <= 0 || Sc <= 0) continue;
            const int q0y = (pad > py) ? (pad - py + stride - 1) / stride : 0;
            const int q0x = (pad > px) ? (pad - px + stride - 1) / stride : 0;
            const int ay = q0y * stride + py - pad, ax = q0x * stride + px - pad;
            if (ay >= H || ax >= W) continue;
            const long long abytes = (long long)K * C * Rc * Sc * 4;
            bool split = use_bf16x3(cfg, C) && abytes < kBufMax;
            int nmax = N;
            if (split && dys * 4 * N >= chunk_limit()) {
                nmax = chunk_samples(dys * 4, 0, cfg.gscale ? cfg.group_n : 1);
                if (nmax <= 0) { split = false; nmax = N; }
            }
            if ((cfg.dy_split || cfg.out_split) && !split) return AG2V_EINVAL;
            // exact arithmetic with >= 128 rows: conv_igemm_f32_ws_kernel (see ag2v_conv2d_fwd_ex)
            static const bool f32ws_on = true;
            bool f32ws = !split && f32ws_on && C >= 128 && K * Rc * Sc >= 64 && abytes < kBufMax;
            if (f32ws && dys * 4 * N >= chunk_limit()) {
                nmax = chunk_samples(dys * 4, 0, cfg.gscale ? cfg.group_n : 1);
                if (nmax <= 0) { f32ws = false; nmax = N; }
            }
            const int cOH = (H - ay + stride - 1) / stride, cOW = (W - ax + stride - 1) / stride;
            if ((long long)nmax * cOH * cOW >= (1LL << 31)) {
                nmax = (int)(((1LL << 31) - 1) / ((long long)cOH * cOW));
                if (cfg.gscale) nmax -= nmax % cfg.group_n;
                if (nmax <= 0) return AG2V_EINVAL;
            }
            const bool tapmaj = igemm_tap_major(K, split);
            if (tapmaj)
                hipLaunchKernelGGL(pack_dgrad_tap_kernel, dim3(ceil_div(C, 64), K / 8), dim3(256), 0, st, w, at, K, C, R, S, py, px,
                                   stride, Rc, Sc);
            else
                hipLaunchKernelGGL(dgrad_weight_prep_kernel, dim3(ceil_div(C, 64), K), dim3(256), 0, st, w, at, K, C, R, S, py, px,
                                   stride, Rc, Sc, split ? 1 : 0);
            auto chunk_args = [&](int n0, int nc) -> ConvArgs {
                float* dxc = dx + (size_t)n0 * dxo;
                ConvArgs f{};
                f.out_nstride = dxo;
                f.tap_major = tapmaj ? 1 : 0;
                f.a_src = at; f.b_src = dy + (size_t)n0 * dys; f.out = dxc;
                f.N = nc; f.C = K; f.H = OH; f.W = OW; f.K = C; f.R = Rc; f.S = Sc; f.stride = 1;
                f.pad = Rc - 1 - q0y; f.pad_w = Sc - 1 - q0x;
                f.OH = cOH; f.OW = cOW;
                f.o_mul = stride; f.o_add_y = ay; f.o_add_x = ax; f.o_H = H; f.o_W = W;
                f.M = C; f.NG = nc * f.OH * f.OW; f.KG = K * Rc * Sc;
                f.a_bytes = (unsigned)abytes; f.b_bytes = (split || f32ws) ? (unsigned)(dys * 4 * nc) : 0u;
                f.f32 = f32ws ? 1 : 0;
                f.gscale = cfg.gscale ? cfg.gscale + n0 / cfg.group_n : nullptr; f.group_n = cfg.group_n;
                f.b_split = cfg.dy_split; f.out_split = cfg.out_split;
                // one class (stride 1): the mask in the epilogue; several classes: one pass at the end
                f.mask = (multi || !mask) ? nullptr : mask + (size_t)n0 * dxs;
                // accumulate: a split-K launch adds onto the existing dx with its atomics (no zeroing); any other launch reads dx
                // back as the epilogue's residual term (each element is read and written by the same thread)
                f.residual = accumulate ? dxc : nullptr;
                f.det = cfg.det;
                return f;
            };
            // channel sums from the epilogue (ag2v_conv_opts::out_chan_sums): one class, written once, and EVERY sample chunk on the kernels
            // that add them -- decided before anything is launched (a launch that adds and one that does not cannot share the buffer)
            bool sums = cfg.chan_sums && !multi && !accumulate && !cfg.det && split;
            for (int n0 = 0; n0 < N && sums; n0 += nmax) {
                ConvArgs t = chunk_args(n0, (N - n0 < nmax) ? N - n0 : nmax);
                sums = halo_chan_sums_ok(t);
            }
            int sums_done = 0, chunks = 0;
            for (int n0 = 0; n0 < N; n0 += nmax) {
                const int nc = (N - n0 < nmax) ? N - n0 : nmax;
                ConvArgs f = chunk_args(n0, nc);
                f.chan_sums = sums ? cfg.chan_sums : nullptr;
                note_chan_sums_done(0);
                const int rc = launch_igemm(f, st, multi || accumulate != 0, split || f32ws);
                if (rc != AG2V_OK) return rc;
                sums_done += chan_sums_done();
                ++chunks;
            }
            if (sums && sums_done != chunks) return AG2V_EINVAL;      // (cannot happen: halo_chan_sums_ok is the launcher's own test)
            note_chan_sums_done(sums ? 1 : 0);
            at += (size_t)K * C * Rc * Sc;
        }
    if (o) { o->kernel = g_last_conv_kernel; o->chan_sums_done = g_chan_sums_done; }
    if (multi && mask) return launch_act_mask(dx, mask, (long long)N * C * H * W, 0, st);
    return AG2V_OK;
}
// The forms without options: process-default arithmetic, no group scale; the <= 4-channel intermediate does not fit the
// C*K*R*S-float workspace these take, so that case runs on the general kernels here (the product passes the full workspace).
extern "C" int ag2v_conv2d_dgrad_acc(const float* dy, const float* w, const float* mask, float* dx,
                                     int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int accumulate,
                                     float* workspace, void* stream) {
    return ag2v_conv2d_dgrad_ex(dy, w, mask, dx, N, C, H, W, K, R, S, stride, pad, accumulate, workspace, (size_t)C * K * R * S,
                                stream, nullptr);
}
extern "C" int ag2v_conv2d_dgrad_masked(const float* dy, const float* w, const float* mask, float* dx,
                                        int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace,
                                        void* stream) {
    return ag2v_conv2d_dgrad_acc(dy, w, mask, dx, N, C, H, W, K, R, S, stride, pad, 0, workspace, stream);
}
extern "C" int ag2v_conv2d_dgrad(const float* dy, const float* w, float* dx,
                                 int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace,
                                 void* stream) {
    return ag2v_conv2d_dgrad_acc(dy, w, nullptr, dx, N, C, H, W, K, R, S, stride, pad, 0, workspace, stream);
}

// ---------------------------------------------------------------- weight gradient of a convolution with <= 4 output channels
// The discriminators' last layer (512 -> 1, 4 x 4, discriminator.py:368-379) and any other few-row layer whose swapped form
// (ops._SWAP_FEW_WGRAD) does not apply: as a GEMM this has 1-4 rows on a 32-row tile and its 130 k-long pixel reduction was gathered
// tap by tap (2 TF, 1.15 ms for 284 MB of input).  Here a workgroup owns ONE input channel (and one slice of the images): a thread
// walks groups of 4 output pixels, keeps the S x (S + 3) input window and the 4 K output gradients in registers and feeds
// 4 K S S FMAs from them; the K S S sums meet by wave reduction + LDS, one atomic add (or, with one slice per channel -- the
// deterministic form -- one plain store) per weight.
template <int KR, int S>
__global__ __launch_bounds__(256) void conv_wgrad_fewrow_kernel(ConvArgs p) {
    __shared__ float red[4][KR * S * S];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int QX = (p.OW + 3) >> 2;
    // blockIdx.z: sample group (ConvArgs::groups consecutive groups of N / groups samples, one gradient each)
    const int ng = p.N / p.groups, grp = blockIdx.z;
    const long long total = (long long)ng * p.OH * QX;
    const long long per = (total + gridDim.y - 1) / gridDim.y;
    const long long q0 = (long long)blockIdx.y * per, q1 = min(total, q0 + per);
    float acc[KR][S][S];
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int r = 0; r < S; ++r)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) acc[k][r][s2] = 0.f;
    const float* __restrict__ dy = p.a_src;
    const float* __restrict__ x = p.b_src;
    const size_t oplane = (size_t)p.OH * p.OW, iplane = (size_t)p.H * p.W;
    for (long long q = q0 + threadIdx.x; q < q1; q += 256) {
        const int xq = (int)(q % QX);
        const long long t = q / QX;
        const int oy = (int)(t % p.OH), n = grp * ng + (int)(t / p.OH);
        const int ox0 = xq * 4;
        float g[KR][4];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const float* gp = dy + ((size_t)n * KR + k) * oplane + (size_t)oy * p.OW + ox0;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[k][j] = (ox0 + j < p.OW) ? gp[j] : 0.f;
        }
        const float* xp = x + ((size_t)n * p.C + c) * iplane;
        const int ix0 = ox0 - p.pad_w;
        float v[S][S + 3];
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const int iy = oy + r - p.pad;
            const bool rowok = (unsigned)iy < (unsigned)p.H;
#pragma unroll
            for (int i = 0; i < S + 3; ++i)
                v[r][i] = (rowok && (unsigned)(ix0 + i) < (unsigned)p.W) ? xp[(size_t)iy * p.W + ix0 + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KR; ++k)
#pragma unroll
            for (int r = 0; r < S; ++r)
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[k][r][s2] = fmaf(g[k][j], v[r][s2 + j], acc[k][r][s2]);
    }
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int r = 0; r < S; ++r)
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                const float tsum = wave_sum(acc[k][r][s2]);
                if (lane == 0) red[wave][(k * S + r) * S + s2] = tsum;
            }
    __syncthreads();
    if (threadIdx.x < KR * S * S) {
        const int i = threadIdx.x, k = i / (S * S), rs = i - k * (S * S);
        const float tsum = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        float* o = p.out + (size_t)grp * p.out_gstride + ((size_t)k * p.C + c) * (S * S) + rs;
        if (gridDim.y > 1) atomicAdd(o, tsum);                 // (the launcher zeroed dw unless it accumulates)
        else *o = p.accumulate ? *o + tsum : tsum;
    }
}
template <int KR>
static bool launch_wgrad_fewrow_k(const ConvArgs& a, dim3 grid, hipStream_t st) {
    switch (a.S) {
        case 1: hipLaunchKernelGGL((conv_wgrad_fewrow_kernel<KR, 1>), grid, dim3(256), 0, st, a); return true;
        case 3: hipLaunchKernelGGL((conv_wgrad_fewrow_kernel<KR, 3>), grid, dim3(256), 0, st, a); return true;
        case 4: hipLaunchKernelGGL((conv_wgrad_fewrow_kernel<KR, 4>), grid, dim3(256), 0, st, a); return true;
        default: return false;
    }
}
static bool wgrad_fewrow_ok(int K, int R, int S, int stride, int groups, int x_split, int dy_split) {
    static const bool on = true;
    return on && K <= 4 && R == S && (S == 1 || S == 3 || S == 4) && stride == 1 && groups >= 1 && groups <= 65535 && !x_split && !dy_split;
}
// det: one slice per channel (plain stores, a fixed summation order)
static int launch_wgrad_fewrow(ConvArgs& a, int det, hipStream_t st) {
    const long long quads = (long long)(a.N / a.groups) * a.OH * ((a.OW + 3) >> 2);
    int slices = det ? 1 : (int)ceil_div(1536, a.C * a.groups);
    const int max_slices = (int)((quads + 2047) / 2048);
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    if (slices > 1 && !a.accumulate) {
        if (hipMemsetAsync(a.out, 0, (size_t)a.groups * a.K * a.C * a.R * a.S * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    const dim3 grid(a.C, slices, a.groups);
    const bool ok = a.K == 1 ? launch_wgrad_fewrow_k<1>(a, grid, st) : a.K == 2 ? launch_wgrad_fewrow_k<2>(a, grid, st)
                  : a.K == 3 ? launch_wgrad_fewrow_k<3>(a, grid, st) : launch_wgrad_fewrow_k<4>(a, grid, st);
    if (!ok) return AG2V_EINVAL;
    set_last_conv_kernel("conv_wgrad_fewrow_kernel");
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// Weight gradient.  accumulate != 0: dw += instead of dw = : a weight used by several forward calls of one step collects its
// gradient in one buffer.  groups > 1 (N % groups == 0): the batch is `groups` consecutive sample groups and dw is
// [groups][K, C, R, S] -- one gradient per group (the frame-batched generator needs dL/dWbar_t per frame t for spectral norm).
// det: deterministic reduction -- a launch whose pixel range is split (spg > 1) stores one partial weight gradient per split into
// det_ws (groups * spg * K*C*R*S floats) and launch_wgrad_det_reduce adds them in order; need (nullable): only plan the launch and
// report that float count (0 when the launch does not split).
static int wgrad_launch(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S, int OH, int OW,
                        int stride, int pad, int accumulate, int groups, bool split, hipStream_t st, int x_split = 0, int dy_split = 0,
                        int det = 0, float* det_ws = nullptr, size_t det_ws_floats = 0, size_t* need = nullptr, bool f32ws = false) {
    ConvArgs a{};
    a.a_src = dy; a.b_src = x; a.out = dw; a.accumulate = accumulate ? 1 : 0;
    a.b_split = x_split; a.a_split = dy_split;
    if ((x_split || dy_split) && !split) return AG2V_EINVAL;
    a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.pad_w = pad;
    a.OH = OH; a.OW = OW;
    const long long Q = (long long)N * OH * OW;
    a.M = K; a.NG = C * R * S; a.KG = (int)Q;
    a.groups = groups; a.gq = (int)(Q / groups); a.out_gstride = (long long)a.M * a.NG;
    const long long mn = (long long)a.M * a.NG;
    int bm = 0;
    if (!split && wgrad_fewrow_ok(K, R, S, stride, groups, x_split, dy_split)) {      // <= 4 rows: one workgroup per input channel
        if (need) { *need = 0; return AG2V_OK; }
        return launch_wgrad_fewrow(a, det, st);
    }
    if (!split && f32ws) {       // exact fp32, 8 + 8 waves, 16-byte loads: the planning of the split kernels
        a.a_bytes = (unsigned)((long long)N * K * OH * OW * 4); a.b_bytes = (unsigned)((long long)N * C * H * W * 4);
        a.f32 = 1;
        if (wgrad_f32_ws_ok(a)) { split = true; plan_wgrad_bf16x3(a); }
        else { a.f32 = 0; a.a_bytes = a.b_bytes = 0; }
    } else if (split) {
        a.a_bytes = (unsigned)((long long)N * K * OH * OW * 4); a.b_bytes = (unsigned)((long long)N * C * H * W * 4);
        plan_wgrad_bf16x3(a);
    }
    if (!split) {
        bm = (a.M <= 32) ? 32 : (a.M <= 64 ? 64 : 128);
        a.tiles_m = ceil_div(a.M, bm);
        a.tiles_n = ceil_div(a.NG, 128);
        const int tiles = a.tiles_m * a.tiles_n;
        // split each group's pixel reduction so that ~1536 workgroups are in flight (256 CUs x up to 4 resident + slack)
        const long long Qg = a.gq;
        int spg = ceil_div(1536, (long long)tiles * groups);
        const int max_split = (int)((Qg + 511) / 512);
        if (spg > max_split) spg = max_split;
        if (spg < 1) spg = 1;
        int kchunk = (int)((Qg + spg - 1) / spg);
        kchunk = ((kchunk + 15) / 16) * 16;
        spg = (int)((Qg + kchunk - 1) / kchunk);
        a.spg = spg; a.splitk = spg * groups; a.kchunk = kchunk;
    }
    // The pixel-range splits of a launch meet either in fp32 atomics on dW or -- deterministic mode, and ALWAYS for the LDS-DMA kernel
    // (conv_wgrad_halo.hip) when the caller provides the workspace -- as slabs that an ordered pass adds: that kernel's workgroups end
    // together and their 74 k atomics per workgroup onto the same tile cost 6-16 % of the launch (tools/wgrad_halo_bench.py, WGH_DET).
    const bool halo = split && !a.f32 && wgrad_halo_ok(a);
    bool slabs = (det || halo) && a.spg > 1;
    const size_t slab_floats = slabs ? (size_t)groups * a.spg * (size_t)mn : 0;
    if (need) { *need = slab_floats; return AG2V_OK; }
    if (slabs && (!det_ws || det_ws_floats < slab_floats)) {
        if (det) return AG2V_EINVAL;                                         // ag2v_conv2d_wgrad_det_workspace_floats()
        slabs = false;                                                       // no workspace: the atomic form
    }
    if (slabs) {
        a.det = 1;
        a.out = det_ws;
    }
    if (split) {
        const int rc = launch_wgrad_bf16x3(a, st);
        if (rc != AG2V_OK) return rc;
    } else {
        if (a.spg > 1 && !a.accumulate && !a.det) {
            if (hipMemsetAsync(dw, 0, (size_t)a.M * a.NG * groups * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
        }
        dim3 grid(a.tiles_m * a.tiles_n, a.splitk);
        if (bm == 32)
            { set_last_conv_kernel("conv_wgrad_kernel<32, 128, 1, 4>"); hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4>), grid, dim3(256), 0, st, a); }
        else if (bm == 64)
            { set_last_conv_kernel("conv_wgrad_kernel<64, 128, 2, 2>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, a); }
        else
            { set_last_conv_kernel("conv_wgrad_kernel<128, 128, 2, 2>"); hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a); }
        AG2V_CHECK_LAUNCH();
    }
    if (slabs) return launch_wgrad_det_reduce(det_ws, dw, mn, groups, a.spg, accumulate, st);
    return AG2V_OK;
}
// the sample-chunk loop of ag2v_conv2d_wgrad_ex; need != null: plan only, *need = the largest per-launch slab workspace (floats)
static int wgrad_ex_impl(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S,
                         int stride, int pad, int accumulate, int groups, hipStream_t st, const ConvCfg& cfg, size_t* need) {
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
    const long long dys = (long long)K * OH * OW, xs = (long long)C * H * W;
    if (xs >= (1LL << 31) || dys >= (1LL << 31)) return AG2V_EINVAL;
    const int Ng = N / groups;
    bool split = use_bf16x3(cfg, K);
    // samples per launch: whole groups, both operands below 4 GiB for the bf16x3 kernels, pixel index below 2^31
    // (dy at most 2 GiB per launch: the 16-byte-load weight-gradient kernels encode "past the range" as byte offset 2^31)
    int nmax = N;
    // exact fp32 on the 16-byte-load pipeline (conv_wgrad_f32_v_kernel) addresses through 32-bit buffer offsets as well: same chunks
    bool f32ws = !split && K >= 128 && stride == 1 && OW % 8 == 0 && W >= 8 && pad <= 3 && true;
    if ((split || f32ws) && (dys * 8 * N >= chunk_limit() || xs * 4 * N >= chunk_limit())) {
        nmax = chunk_samples(dys * 8, xs * 4, groups > 1 ? Ng : 1);
        if (nmax <= 0) nmax = chunk_samples(dys * 4, xs * 4, groups > 1 ? Ng : 1);
        if (nmax <= 0) { split = false; f32ws = false; nmax = N; }
    }
    if ((long long)nmax * OH * OW >= (1LL << 31)) {
        nmax = (int)(((1LL << 31) - 1) / ((long long)OH * OW));
        if (groups > 1) nmax -= nmax % Ng;
        if (nmax <= 0) return AG2V_EINVAL;
    }
    const size_t wsz = (size_t)K * C * R * S;
    size_t most = 0;
    for (int n0 = 0; n0 < N; n0 += nmax) {
        const int nc = (N - n0 < nmax) ? N - n0 : nmax;
        const int g0 = n0 / Ng, gc = (groups > 1) ? nc / Ng : 1;
        float* out = (groups > 1 && dw) ? dw + (size_t)g0 * wsz : dw;
        const int acc = (groups > 1) ? accumulate : (accumulate || n0 > 0);
        size_t one = 0;
        const int rc = wgrad_launch(need ? nullptr : dy + (size_t)n0 * dys, need ? nullptr : x + (size_t)n0 * xs, out, nc, C, H, W, K, R, S,
                                    OH, OW, stride, pad, acc, gc, split, st, cfg.in_split, cfg.dy_split, cfg.det, cfg.det_ws,
                                    cfg.det_ws_floats, need ? &one : nullptr, f32ws);
        if (rc != AG2V_OK) return rc;
        if (one > most) most = one;
    }
    if (need) *need = most;
    return AG2V_OK;
}
extern "C" int ag2v_conv2d_wgrad_ex(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S,
                                    int stride, int pad, int accumulate, int groups, void* stream, ag2v_conv_opts* o) {
    if (!dy || !x || !dw || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || groups <= 0 || N % groups != 0) return AG2V_EINVAL;
    const ConvCfg cfg = conv_cfg(o);
    const int rc = wgrad_ex_impl(dy, x, dw, N, C, H, W, K, R, S, stride, pad, accumulate, groups, (hipStream_t)stream, cfg, nullptr);
    if (rc == AG2V_OK && o) o->kernel = g_last_conv_kernel;
    return rc;
}
// Floats of ag2v_conv_opts::det_workspace an ag2v_conv2d_wgrad_ex call with these arguments and options (o->deterministic included) can
// use: the slabs of a deterministic call, or of a call that takes the LDS-DMA kernel, which adds its splits through slabs in either mode
// when it is given them (0: atomics or no split, or the arguments are invalid).  Same planning code as the call itself.
extern "C" size_t ag2v_conv2d_wgrad_det_workspace_floats(int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                                                         int groups, const ag2v_conv_opts* o) {
    if (N <= 0 || C <= 0 || K <= 0 || stride <= 0 || groups <= 0 || N % groups != 0) return 0;
    ConvCfg cfg = conv_cfg(o);         // (o->deterministic as the call will pass it: the LDS-DMA kernel wants slabs in either mode)
    size_t need = 0;
    if (wgrad_ex_impl(nullptr, nullptr, nullptr, N, C, H, W, K, R, S, stride, pad, 0, groups, nullptr, cfg, &need) != AG2V_OK) return 0;
    return need;
}
// 1 when a convolution of this shape can take its INPUT x as split words (ag2v_conv_opts::in_split) in the forward pass and in
// the weight gradient with these options: both passes on the bf16x3 kernels (rows >= min_rows, precision), the weight gradient on
// the 16-byte-load kernels (the only ones with a split-word x path).  Callers decide the format of an activation tensor with
// this before they produce it (ag2v_norm_apply_g(out_split)); a call that cannot honour in_split returns AG2V_EINVAL.
extern "C" int ag2v_conv2d_split_input_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows_fwd,
                                          int min_rows_bwd, int precision) {
    if (N <= 0 || C <= 0 || K <= 0 || stride <= 0 || R * S > 16) return 0;
    ag2v_conv_opts of{}; of.precision = precision; of.min_rows = min_rows_fwd;
    ag2v_conv_opts ob{}; ob.precision = precision; ob.min_rows = min_rows_bwd;
    const ConvCfg cf = conv_cfg(&of), cb = conv_cfg(&ob);
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return 0;
    const int KG = C * R * S;
    const long long abytes = (long long)KG * K * 4, xs = (long long)C * H * W, dys = (long long)K * OH * OW;
    static const bool short_exact = true;
    if (!(use_bf16x3(cf, K) && abytes < kBufMax && !(short_exact && KG < 64 && K < 128))) return 0;        // forward (ag2v_conv2d_fwd_ex)
    if (xs * 4 > chunk_limit()) return 0;                                                                   // one sample must fit a launch
    if (!use_bf16x3(cb, K)) return 0;                                                                      // weight gradient
    if (dys * 8 > chunk_limit() && dys * 4 > chunk_limit()) return 0;
    if (!(stride == 1 && OW % 8 == 0 && W >= 8 && pad <= 3) || false) return 0;
    return 1;
}
// The same question for the OUTPUT gradient dy of the convolution (ag2v_conv_opts::dy_split): its data gradient (dy is the
// gathered operand there) and its weight gradient both on the bf16x3 kernels, the latter on the 16-byte-load form.
extern "C" int ag2v_conv2d_split_dy_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows_bwd,
                                       int precision) {
    if (N <= 0 || C <= 0 || K <= 0 || stride != 1 || R * S > 16) return 0;
    ag2v_conv_opts ob{}; ob.precision = precision; ob.min_rows = min_rows_bwd;
    const ConvCfg cb = conv_cfg(&ob);
    const int OH = conv_out_dim(H, R, stride, pad), OW = conv_out_dim(W, S, stride, pad);
    if (OH <= 0 || OW <= 0) return 0;
    const long long dys = (long long)K * OH * OW;
    if (few_channel_dgrad(N, C, K, R, S, OH, OW, stride, false)) return 0;
    if (!use_bf16x3(cb, C) || (long long)K * C * R * S * 4 >= kBufMax) return 0;          // data gradient: rows = input channels
    if (!use_bf16x3(cb, K)) return 0;                                                       // weight gradient: rows = output channels
    if (dys * 8 > chunk_limit()) return 0;
    if (!(OW % 8 == 0 && W >= 8 && pad <= 3) || false) return 0;
    return 1;
}
extern "C" int ag2v_conv2d_wgrad_acc(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S,
                                     int stride, int pad, int accumulate, void* stream) {
    return ag2v_conv2d_wgrad_ex(dy, x, dw, N, C, H, W, K, R, S, stride, pad, accumulate, 1, stream, nullptr);
}
extern "C" int ag2v_conv2d_wgrad(const float* dy, const float* x, float* dw,
                                 int N, int C, int H, int W, int K, int R, int S, int stride, int pad, void* stream) {
    return ag2v_conv2d_wgrad_ex(dy, x, dw, N, C, H, W, K, R, S, stride, pad, 0, 1, stream, nullptr);
}
