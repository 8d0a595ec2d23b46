// Implicit-GEMM 2-D convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, 157.3 TF peak).
//
// Replaces every nn.Conv2d call site of the reference hot path:
//   generator 3x3 / 1x1 convs   models/spade_models/networks/{generator.py:30, flows_generator.py:32-54,82-85,
//                               spade_generator.py:16-34, architecture.py:29-32, normalization.py:90-94}
//   discriminator 4x4 s2/s1 p2  models/spade_models/networks/discriminator.py:368-379
// forward, data gradient and weight gradient (what autograd derives from aten.convolution there).
//
// GEMM views (NCHW fp32, weights [K,C,R,S]):
//   FWD   Y[k,(n,oy,ox)]  = sum_{c,r,s} Wt[k,(c,r,s)] * X[n,c,oy*st+r-pad,ox*st+s-pad]
//   DGRAD dX[c,(n,y,x)]   = sum_{k,r,s} Wt[k,c,r,s]   * dY[n,k,(y+pad-r)/st,(x+pad-s)/st]   (divisible taps only)
//   WGRAD dW[k,(c,r,s)]   = sum_{n,oy,ox} dY[n,k,oy,ox] * X[n,c,oy*st+r-pad,ox*st+s-pad]     (split over pixels)
// The B operand is gathered straight from the NCHW tensor with lanes along the pixel index (coalesced rows),
// staged through LDS as [k][m]/[k][n] so every MFMA operand read is a conflict-free ds_read_b32, and the next
// K-tile's global loads are in flight while the current tile's 32 MFMAs per wave issue.
#include "common.h"

struct ConvArgs {
    const float* a_src;     // weights (FWD/DGRAD) | dy (WGRAD)
    const float* b_src;     // x (FWD) | dy (DGRAD) | x (WGRAD)
    float* out;             // y | dx | dw
    const float* bias;      // FWD: [K] or null
    const float* residual;  // FWD: same shape as y, or null
    int N, C, H, W;         // input image
    int K, R, S, OH, OW;    // filters / output image
    int stride, pad;        // pad = padding along y
    int pad_w;              // padding along x (== pad except for the parity sub-filters of a strided dgrad)
    int o_mul, o_add_y, o_add_x, o_H, o_W;  // FWD output placement: (oy,ox) -> (oy*o_mul+o_add_y, ox*o_mul+o_add_x) in [o_H,o_W]
    int M, NG, KG;          // GEMM dims of this mode
    int tiles_m, tiles_n;
    int splitk, kchunk;     // WGRAD: number of pixel-range splits, pixels per split (multiple of 16)
                            // FWD/DGRAD: number of K-range splits (grid.y), K-tiles per split; >1 => atomic epilogue
};

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// Division by the filter footprint: 1x1, 3x3 and 4x4 (the only shapes on the path) get constant divisors.
__device__ __forceinline__ int div_small(int v, int d) {
    if (d == 9) return v / 9;
    if (d == 16) return v >> 4;
    if (d == 1) return v;
    if (d == 3) return v / 3;
    if (d == 4) return v >> 2;
    return v / d;
}
__device__ __forceinline__ void decode_crs(int kg, int RS, int S, int& c, int& r, int& s) {
    c = div_small(kg, RS);
    const int rs = kg - c * RS;
    r = div_small(rs, S);
    s = rs - r * S;
}

template <int MODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BK = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int EA = BM * BK / 256;   // A elements per thread: EA consecutive k of one row
    constexpr int TPR = BK / EA;        // threads per A row
    constexpr int EB = BK * BN / 256;   // B elements per thread: one pixel column, EB k-rows
    constexpr int KSB = 256 / BN;       // k-row step between a thread's B elements
    static_assert(TM >= 1 && TN >= 1, "tile");
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;  // m-tiles of one pixel tile are neighbours
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int wm0 = (wave / WN) * (TM * 32), wn0 = (wave % WN) * (TN * 32);
    const int RS = p.R * p.S;

    // ---- B gather state: this thread owns pixel column jb of the tile for the whole K loop
    const int jb = tid % BN, kkb0 = tid / BN;
    const int PH = (MODE == MODE_FWD) ? p.OH : p.H, PW = (MODE == MODE_FWD) ? p.OW : p.W;
    const int pix = n0 + jb;
    const bool pvalid = pix < p.NG;
    const int n_img = pix / (PH * PW);
    const int prem = pix - n_img * (PH * PW);
    const int py = prem / PW, px = prem - py * PW;
    // Source plane geometry: FWD gathers x[N,C,H,W]; DGRAD gathers dy[N,K,OH,OW].
    const int SH = (MODE == MODE_FWD) ? p.H : p.OH, SW = (MODE == MODE_FWD) ? p.W : p.OW;
    const int SC = (MODE == MODE_FWD) ? p.C : p.K;
    const int SHW = SH * SW;
    // Per-pixel tap table: bit (r*S+s) of vmask says tap (r,s) reads inside the source plane; the tap's address is
    // bptr + chan*SHW + toff(r,s) with a wave-uniform (scalar) chan/tap part.  FWD (and stride-1 DGRAD) taps are affine
    // in (r,s); a strided DGRAD tap ((y+pad-r)/st) is not, so it keeps a per-lane offset computed per element.
    const float* bptr = p.b_src + (size_t)n_img * SC * SHW;
    unsigned vmask = 0;
    int y0, x0;
    if (MODE == MODE_FWD) {
        y0 = py * p.stride - p.pad;
        x0 = px * p.stride - p.pad_w;
    } else {
        y0 = py + p.pad;
        x0 = px + p.pad_w;
    }
    if (pvalid) {
        for (int r = 0; r < p.R; ++r)
            for (int s2 = 0; s2 < p.S; ++s2) {
                bool ok;
                if (MODE == MODE_FWD) {
                    ok = (unsigned)(y0 + r) < (unsigned)SH && (unsigned)(x0 + s2) < (unsigned)SW;
                } else {
                    const int ty = y0 - r, tx = x0 - s2;
                    ok = ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0 && (ty / p.stride) < SH &&
                         (tx / p.stride) < SW;
                }
                if (ok) vmask |= 1u << (r * p.S + s2);
            }
    }
    const bool affine = (MODE == MODE_FWD) || p.stride == 1;
    if (affine) bptr += (MODE == MODE_FWD) ? (y0 * SW + x0) : (y0 * SW + x0);   // tap (0,0); may point before the plane
    // ---- A load state
    const int ma = tid / TPR, kka0 = (tid % TPR) * EA;
    const bool mvalid = (m0 + ma) < p.M;
    const float* aptr = p.a_src + ((MODE == MODE_FWD) ? (size_t)(m0 + ma) * p.KG : (size_t)(m0 + ma) * RS);

    // K-index decode table: the 16 k-rows of a K-tile are decoded ONCE per workgroup (16 lanes in parallel) into LDS as
    // {offset of (chan,tap) inside a source image, tap bit | r<<16 | s<<20}; every gathered element then costs one
    // broadcast ds_read_b64, a mask test and one 64-bit add.  (A per-element scalar decode cost ~10 SALU per MFMA.)
    __shared__ int2 tab[2][BK];
    auto build_tab = [&](int kt) {
        if (tid < BK) {
            const int kg = kt * BK + tid;
            int c, r, s2;
            decode_crs(kg, RS, p.S, c, r, s2);
            int koff;
            if (MODE == MODE_FWD) koff = c * SHW + r * SW + s2;
            else if (affine) koff = c * SHW - r * SW - s2;
            else koff = c * SHW;
            const unsigned bit = (kg < p.KG) ? (1u << (r * p.S + s2)) : 0u;
            tab[kt & 1][tid] = make_int2(koff, (int)(bit | ((unsigned)r << 16) | ((unsigned)s2 << 20)));
        }
    };
    auto fetchB = [&](int kt, int kk) -> float {
        const int2 e = tab[kt & 1][kk];
        const bool ok = (vmask & (unsigned)e.y & 0xFFFFu) != 0u;
        if (MODE == MODE_FWD || affine) {
            return ok ? bptr[e.x] : 0.f;
        } else {
            const int r = (e.y >> 16) & 15, s2 = (e.y >> 20) & 15;
            const int oy = (y0 - r) / p.stride, ox = (x0 - s2) / p.stride;
            return ok ? bptr[e.x + oy * SW + ox] : 0.f;
        }
    };
    auto fetchA = [&](int kg) -> float {
        const bool ok = mvalid && kg < p.KG;
        if (MODE == MODE_FWD) {
            return ok ? aptr[kg] : 0.f;
        } else {
            const int k = div_small(kg, RS);
            const int rs = kg - k * RS;
            return ok ? aptr[(size_t)k * p.C * RS + rs] : 0.f;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[EA], rb[EB];
    const int nk_all = (p.KG + BK - 1) / BK;
    const int kt0 = (p.splitk > 1) ? blockIdx.y * p.kchunk : 0;          // small-N layers: split the K loop over grid.y
    const int nk = (p.splitk > 1) ? min(nk_all, kt0 + p.kchunk) : nk_all;
    if (kt0 >= nk) return;  // uniform per workgroup
    build_tab(kt0);
    __syncthreads();
    build_tab(kt0 + 1);
#pragma unroll
    for (int i = 0; i < EA; ++i) ra[i] = fetchA(kt0 * BK + kka0 + i);
#pragma unroll
    for (int i = 0; i < EB; ++i) rb[i] = fetchB(kt0, kkb0 + KSB * i);
#pragma unroll
    for (int i = 0; i < EA; ++i) As[0][kka0 + i][ma] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[0][kkb0 + KSB * i][jb] = rb[i];
    __syncthreads();

    int cur = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            const int k0 = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < EA; ++i) ra[i] = fetchA(k0 + kka0 + i);
#pragma unroll
            for (int i = 0; i < EB; ++i) rb[i] = fetchB(kt + 1, kkb0 + KSB * i);
            build_tab(kt + 2);   // tab[kt&1] was last read while fetching tile kt, i.e. before the previous barrier
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
            for (int i = 0; i < EA; ++i) As[cur ^ 1][kka0 + i][ma] = ra[i];
#pragma unroll
            for (int i = 0; i < EB; ++i) Bs[cur ^ 1][kkb0 + KSB * i][jb] = rb[i];
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int OUTC = (MODE == MODE_FWD) ? p.K : p.C;
    const int gplane = PH * PW;                                    // pixels of the GEMM's pixel space per image
    const int plane = (MODE == MODE_FWD) ? p.o_H * p.o_W : gplane;  // elements of one output channel plane
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int opix = n0 + wn0 + j * 32 + (lane & 31);
        if (opix >= p.NG) continue;
        const int on = opix / gplane;
        int orem = opix - on * gplane;
        if (MODE == MODE_FWD && (p.o_mul != 1 || p.o_add_y != 0 || p.o_add_x != 0 || p.o_W != PW)) {
            const int qy = orem / PW, qx = orem - qy * PW;
            const int yy = qy * p.o_mul + p.o_add_y, xx = qx * p.o_mul + p.o_add_x;
            if ((unsigned)yy >= (unsigned)p.o_H || (unsigned)xx >= (unsigned)p.o_W) continue;
            orem = yy * p.o_W + xx;
        }
        float* obase = p.out + (size_t)on * OUTC * plane + orem;
        const float* rbase = (MODE == MODE_FWD && p.residual) ? p.residual + (size_t)on * OUTC * plane + orem : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float v = acc[i][j][r];
                    if (MODE == MODE_FWD && kt0 == 0) {
                        if (p.bias) v += p.bias[m];
                        if (rbase) v += rbase[(size_t)m * plane];
                    }
                    if (p.splitk > 1)
                        atomicAdd(obase + (size_t)m * plane, v);
                    else
                        obase[(size_t)m * plane] = v;
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
    const int Q = p.N * OHW;
    const int qs = blockIdx.y * p.kchunk;
    const int qe = min(qs + p.kchunk, Q);
    if (qs >= qe) return;  // uniform per block

    const int qq = tid & 15, rg = tid >> 4;
    // hoisted column decode: j -> (c, r-pad, s-pad)
    int boff[EB], bdr[EB], bds[EB];
#pragma unroll
    for (int i = 0; i < EB; ++i) {
        const int j = n0 + rg + 16 * i;
        int c, r, s;
        decode_crs(j, RS, p.S, c, r, s);
        bdr[i] = (j < p.NG) ? (r - p.pad) : (1 << 28);  // invalid column => always out of bounds
        bds[i] = s - p.pad;
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
            const int iy = yb + bdr[i], ix = xb + bds[i];
            const bool ok = qvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
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
                    float* dst = p.out + (size_t)m * p.NG + col;
                    if (p.splitk > 1)
                        atomicAdd(dst, acc[i][j][r]);
                    else
                        *dst = acc[i][j][r];
                }
            }
        }
    }
}

// wt[c][k][r][s] = w[k][c][R-1-r][S-1-s]: with it a stride-1 data gradient is a plain forward convolution of dy
// (pad' = R-1-pad), so it runs on the forward kernel with contiguous weight rows.
__global__ __launch_bounds__(256) void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K,
                                                                    int C, int RS) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)K * C * RS) return;
    const int rs = (int)(i % RS);
    const long long t = i / RS;
    const int k = (int)(t % K), c = (int)(t / K);
    wt[i] = w[((size_t)k * C + c) * RS + (RS - 1 - rs)];
}

// Sub-filter of parity class (py,px) of a stride-2 data gradient, flipped and transposed for the forward kernel:
//   wt[c][k][r''][s''] = w[k][c][py + 2*(Rc-1-r'')][px + 2*(Sc-1-s'')]
__global__ __launch_bounds__(256) void weight_parity_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int C,
                                                                 int R, int S, int py, int px, int Rc, int Sc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)K * C * Rc * Sc) return;
    const int s2 = (int)(i % Sc);
    long long t = i / Sc;
    const int r2 = (int)(t % Rc);
    t /= Rc;
    const int k = (int)(t % K), c = (int)(t / K);
    const int r = py + 2 * (Rc - 1 - r2), s = px + 2 * (Sc - 1 - s2);
    wt[i] = w[(((size_t)k * C + c) * R + r) * S + s];
}

static int conv_out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

template <int MODE>
static int launch_igemm(ConvArgs& a, hipStream_t st, bool out_zeroed = false) {
    if (a.M <= 0 || a.NG <= 0 || a.KG <= 0) return AG2V_EINVAL;
    if (a.o_mul == 0) { a.o_mul = 1; a.o_add_y = 0; a.o_add_x = 0; a.o_H = a.OH; a.o_W = a.OW; }
    const int bm = (a.M <= 32) ? 32 : (a.M <= 64 ? 64 : 128);
    const int bn = (bm == 128) ? 128 : 256;
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = ceil_div(a.NG, bn);
    // Layers with few output pixels (8x8 ... 32x32 maps) have fewer tiles than the chip has CUs while their K loop is
    // thousands of steps long: split K over grid.y and accumulate with fp32 atomics into the zeroed output.
    const int tiles = a.tiles_m * a.tiles_n, nk = ceil_div(a.KG, 16);
    a.splitk = 1; a.kchunk = nk;
    if (tiles < 384 && nk >= 16) {
        int sk = ceil_div(768, tiles);
        if (sk > nk / 8) sk = nk / 8;
        if (sk > 1) {
            a.kchunk = ceil_div(nk, sk);
            a.splitk = ceil_div(nk, a.kchunk);
        }
    }
    if (a.splitk > 1 && !out_zeroed) {
        const size_t out_elems = (MODE == MODE_FWD) ? (size_t)a.N * a.K * a.o_H * a.o_W : (size_t)a.N * a.C * a.H * a.W;
        if (hipMemsetAsync(a.out, 0, out_elems * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    dim3 grid(tiles, a.splitk);
    if (bm == 32)
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 32, 256, 1, 4>), grid, dim3(256), 0, st, a);
    else if (bm == 64)
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 64, 256, 1, 4>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<MODE, 128, 128, 2, 2>), grid, dim3(256), 0, st, a);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" int ag2v_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                               int N, int C, int H, int W, int K, int R, int S, int stride, int pad, void* stream) {
    if (!x || !w || !y || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || R * S > 16) return AG2V_EINVAL;
    ConvArgs a{};
    a.a_src = w; a.b_src = x; a.out = y; a.bias = bias; a.residual = residual;
    a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.pad_w = pad;
    a.OH = conv_out_dim(H, R, stride, pad); a.OW = conv_out_dim(W, S, stride, pad);
    if (a.OH <= 0 || a.OW <= 0) return AG2V_EINVAL;
    if ((long long)N * a.OH * a.OW >= (1LL << 31) || (long long)C * R * S >= (1LL << 31)) return AG2V_EINVAL;
    a.M = K; a.NG = N * a.OH * a.OW; a.KG = C * R * S;
    return launch_igemm<MODE_FWD>(a, (hipStream_t)stream);
}

extern "C" size_t ag2v_conv2d_dgrad_workspace_floats(int C, int K, int R, int S, int stride) {
    return (stride == 1 || stride == 2) ? (size_t)C * K * R * S : 0;
}

extern "C" int ag2v_conv2d_dgrad(const float* dy, const float* w, float* dx,
                                 int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace,
                                 void* stream) {
    if (!dy || !w || !dx || N <= 0 || C <= 0 || K <= 0 || stride <= 0 || R * S > 16) return AG2V_EINVAL;
    if (stride == 1 && workspace && pad <= R - 1 && pad <= S - 1 && R == S) {
        // dX = conv_fwd(dY, flip-transpose(W), pad' = R-1-pad)
        hipStream_t st = (hipStream_t)stream;
        const long long nw = (long long)K * C * R * S;
        hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3(ceil_div(nw, 256)), dim3(256), 0, st, w, workspace, K, C, R * S);
        const int OH = conv_out_dim(H, R, 1, pad), OW = conv_out_dim(W, S, 1, pad);
        if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
        ConvArgs f{};
        f.a_src = workspace; f.b_src = dy; f.out = dx;
        f.N = N; f.C = K; f.H = OH; f.W = OW; f.K = C; f.R = R; f.S = S; f.stride = 1; f.pad = R - 1 - pad; f.pad_w = S - 1 - pad;
        f.OH = H; f.OW = W;
        if ((long long)N * H * W >= (1LL << 31) || (long long)K * R * S >= (1LL << 31)) return AG2V_EINVAL;
        f.M = C; f.NG = N * H * W; f.KG = K * R * S;
        return launch_igemm<MODE_FWD>(f, st);
    }
    if (stride == 2 && workspace) {
        // Stride-2 data gradient by output-pixel parity: pixels with (y+pad)%2 == py, (x+pad)%2 == px only see the taps
        // r = py + 2r', s = px + 2s', i.e. each class is a stride-1 correlation of dY with a (R/2 x S/2) sub-filter
        // -> four forward-kernel launches with 1/4 of the taps each instead of one launch that multiplies 3/4 zeros.
        hipStream_t st = (hipStream_t)stream;
        const int OH = conv_out_dim(H, R, 2, pad), OW = conv_out_dim(W, S, 2, pad);
        if (OH <= 0 || OW <= 0) return AG2V_EINVAL;
        if (hipMemsetAsync(dx, 0, (size_t)N * C * H * W * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
        float* wt = workspace;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int Rc = (R - py + 1) / 2, Sc = (S - px + 1) / 2;   // taps r in [0,R) with r%2 == py
                if (Rc <= 0 || Sc <= 0) continue;
                const long long nw = (long long)K * C * Rc * Sc;
                hipLaunchKernelGGL(weight_parity_flip_kernel, dim3(ceil_div(nw, 256)), dim3(256), 0, st, w, wt, K, C, R, S, py, px, Rc, Sc);
                ConvArgs f{};
                f.a_src = wt; f.b_src = dy; f.out = dx;
                f.N = N; f.C = K; f.H = OH; f.W = OW; f.K = C; f.R = Rc; f.S = Sc; f.stride = 1; f.pad = Rc - 1; f.pad_w = Sc - 1;
                f.OH = OH + Rc - 1; f.OW = OW + Sc - 1;
                f.o_mul = 2; f.o_add_y = py - pad; f.o_add_x = px - pad; f.o_H = H; f.o_W = W;
                if ((long long)N * f.OH * f.OW >= (1LL << 31)) return AG2V_EINVAL;
                f.M = C; f.NG = N * f.OH * f.OW; f.KG = K * Rc * Sc;
                const int rc = launch_igemm<MODE_FWD>(f, st, true);
                if (rc != AG2V_OK) return rc;
                wt += nw;
            }
        return AG2V_OK;
    }
    ConvArgs a{};
    a.a_src = w; a.b_src = dy; a.out = dx;
    a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.pad_w = pad;
    a.OH = conv_out_dim(H, R, stride, pad); a.OW = conv_out_dim(W, S, stride, pad);
    if (a.OH <= 0 || a.OW <= 0) return AG2V_EINVAL;
    if ((long long)N * H * W >= (1LL << 31) || (long long)K * R * S >= (1LL << 31)) return AG2V_EINVAL;
    a.M = C; a.NG = N * H * W; a.KG = K * R * S;
    return launch_igemm<MODE_DGRAD>(a, (hipStream_t)stream);
}

extern "C" int ag2v_conv2d_wgrad(const float* dy, const float* x, float* dw,
                                 int N, int C, int H, int W, int K, int R, int S, int stride, int pad, void* stream) {
    if (!dy || !x || !dw || N <= 0 || C <= 0 || K <= 0 || stride <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a{};
    a.a_src = dy; a.b_src = x; a.out = dw;
    a.N = N; a.C = C; a.H = H; a.W = W; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.pad_w = pad;
    a.OH = conv_out_dim(H, R, stride, pad); a.OW = conv_out_dim(W, S, stride, pad);
    if (a.OH <= 0 || a.OW <= 0) return AG2V_EINVAL;
    const long long Q = (long long)N * a.OH * a.OW;
    if (Q >= (1LL << 31) || (long long)C * H * W >= (1LL << 31)) return AG2V_EINVAL;
    a.M = K; a.NG = C * R * S; a.KG = (int)Q;
    int bm = (a.M <= 32) ? 32 : (a.M <= 64 ? 64 : 128);
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = ceil_div(a.NG, 128);
    const int tiles = a.tiles_m * a.tiles_n;
    // split the pixel reduction so that ~1536 workgroups are in flight (256 CUs x up to 4 resident + slack)
    int splitk = ceil_div(1536, tiles);
    const int max_split = (int)((Q + 511) / 512);
    if (splitk > max_split) splitk = max_split;
    if (splitk < 1) splitk = 1;
    int kchunk = (int)((Q + splitk - 1) / splitk);
    kchunk = ((kchunk + 15) / 16) * 16;
    splitk = (int)((Q + kchunk - 1) / kchunk);
    a.splitk = splitk; a.kchunk = kchunk;
    if (splitk > 1) {
        if (hipMemsetAsync(dw, 0, (size_t)a.M * a.NG * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    dim3 grid(tiles, splitk);
    if (bm == 32)
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4>), grid, dim3(256), 0, st, a);
    else if (bm == 64)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
