// Convolution on the bf16 matrix pipe with fp32-level accuracy ("bf16x3"), gfx950.
//
// Every fp32 operand is split as x = hi + lo (+O(2^-18|x|)), hi = bf16(x), lo = bf16(x - hi); a product is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 / 16x16x32 (exact 16-bit products, fp32 accumulation).  The dropped
// a_lo*b_lo term and the split residuals are <= ~1e-5 relative -- two orders inside the 1e-3 parity bound of the step and
// inside the 2e-4 bound of the kernel tests (tests/test_ops_gpu.py runs every conv case in both arithmetic modes) -- while the
// bf16 pipe runs 16x the fp32 MFMA rate, i.e. 16/3 of it per fp32-equivalent product.
//
// Structure (all kernels): one workgroup = 16 waves on one CU: 8 PRODUCER waves (global loads two K-tiles ahead, split, LDS
// stores) and 8 CONSUMER waves (a 64x64 sub-tile each: fragment reads and MFMAs only); block tile 256x128 or 128x256, K-tile 32;
// LDS holds the hi and lo images of both operands as k-contiguous rows of 32 bf16 (+8 pad = 80 B: ds_read_b128 /
// ds_write_b128 conflict-free), double-buffered; one workgroup barrier per K-tile.  The weight operand of the forward / data-
// gradient kernel arrives pre-split (packed hi:lo words, k-major) from the prep kernels, so its in-kernel cost is two v_perm
// per pair; activations are split in-kernel (v_cvt_pk_bf16_f32).  History and measurements: DESIGN.md 3.4.
#include <type_traits>
#include "conv_common.h"
#include <stdlib.h>

// Forward / data-gradient implicit GEMM.  A homogeneous first version (8 waves that all loaded, split and multiplied: 7 VALU
// per MFMA, the MFMA pipe 32 % busy, waves 51 % in s_waitcnt / barrier because every wave converts at the same time) is in the
// history; here the two jobs sit on different waves of each SIMD, which the hardware runs concurrently (separate pipes), and
// one workgroup barrier per K-tile hands the buffers over.
// pair of split words -> (hi pair, lo pair) LDS words
__device__ __forceinline__ void regroup2(float w0, float w1, unsigned& hi, unsigned& lo) {
    hi = __builtin_amdgcn_perm(__float_as_uint(w1), __float_as_uint(w0), 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(w1), __float_as_uint(w0), 0x05040100u);
}
template <bool PS, bool H16 = false>
__device__ __forceinline__ void split_or_regroup(float x0, float x1, unsigned& hi, unsigned& lo) {
    if constexpr (PS) regroup2(x0, x1, hi, lo);
    else if constexpr (H16) split2_h(x0, x1, hi, lo);
    else split2(x0, x1, hi, lo);
}
// the three MFMAs of one 32x32x16 product block on bf16 or fp16 operand words (same registers: eight 16-bit values per lane)
template <bool H16>
__device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    if constexpr (H16) {
        const f16x8 xah = __builtin_bit_cast(f16x8, ah), xal = __builtin_bit_cast(f16x8, al);
        const f16x8 xbh = __builtin_bit_cast(f16x8, bh), xbl = __builtin_bit_cast(f16x8, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xal, xbh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xah, xbh, acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    return acc;
}
#ifndef AG2V_LDK
#define AG2V_LDK 40      // LDS row stride in bf16 elements (tools/ubench builds override it for layout experiments)
#endif
#ifndef AG2V_TALL_MF16
#define AG2V_TALL_MF16 false     // (true: 16x16x32 MFMAs on the table-path 256 x 128 tile; its epilogue is the unbatched one)
#endif
#ifndef AG2V_WIDE_MF16
#define AG2V_WIDE_MF16 false
#endif
#ifndef AG2V_CONSUMER_SKEW
#define AG2V_CONSUMER_SKEW 1      // 0: the unskewed consumer loop (A/B builds)
#endif
// PSB: the gathered tensor holds SPLIT WORDS already -- (bf16 hi << 16) | bf16 lo of each value, written by the kernel that
// produced it (ag2v_norm_apply_g(out_split), pack_split()); the producers then only regroup halves (2 v_perm per pair instead
// of the 6-instruction split).  Same products, same sums: results are bit-identical to the fp32-input kernel's.
// F32: the same wave-specialised pipeline on EXACT fp32 operands (v_mfma_f32_32x32x2_f32): the producers store the fp32 words as
// loaded, [row][k] with 36-word rows; consumer lane (row, half) multiplies k = 16 half + j, j < 16 (the order of the k terms of a
// K-tile is free as long as both operands use the same), i.e. four 16-byte LDS reads per 32-row fragment.  Table path only.
// H16: fp16 3-term split instead of bf16 (split2_h, v_mfma_f32_32x32x16_f16; weights packed by the *_h pack kernels): 32x32x16 tiles,
// fp32 gathered operand.
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16, bool TAPK, bool B4, bool PSB, bool F32 = false, bool H16 = false>
__device__ __forceinline__ void igemm_bf16x3_body(const ConvArgs& p) {
    static_assert(!H16 || (!MF16 && !PSB && !F32), "fp16 split: 32x32x16 tiles on fp32 inputs");
    constexpr int BK = 32, LDK = AG2V_LDK, NP = NPW * 64;       // NP producer threads
    constexpr int TM = BM / CWM / 32, TN = BN / CWN / 32;  // consumer wave tile in 32x32 MFMA tiles
    constexpr int AP = NP / BM > 0 ? NP / BM : 1, AK = BK / AP;
    constexpr int BP = NP / BN > 0 ? NP / BN : 1, BKK = BK / BP;
    constexpr int ACOLS = BM / (NP / AP), BCOLS = BN / (NP / BP);   // columns per producer thread (1 or 2 when the tile side is 256)
    static_assert(BM % (CWM * 32) == 0 && BN % (CWN * 32) == 0, "consumer wave tiles");
    static_assert(AK % 8 == 0 && BKK % 8 == 0, "16-byte LDS stores");
    // B4 (tap-major, stride 1, OW % 4 == 0): a producer thread owns a QUAD of 4 consecutive pixels and KPT channels of the K-tile
    // and fetches each channel with one 16-byte load (a dwordx4 wave-load costs the texture addresser ~20 cycles, four dword
    // ones ~40: the gathered operand of the 128 x 256 tile was 16 of a thread's 18 loads).  Pixel p lives in LDS row
    // (p & 3) * GR + (p >> 2): a wave's stores for quad element j go to consecutive rows, and GR = BN/4 + 4 keeps the consumers'
    // 16-lane fragment reads on distinct banks (rows 4 g + t, g < 4, t < 4 of an 80-byte pitch).
    constexpr int QN = BN / 4, KP = NP / QN, KPT = BK / KP, GR = QN + 4;
    constexpr int BROWS = B4 ? 4 * GR : BN;
    static_assert(!B4 || (TAPK && BCOLS == 1 && 4 * KPT == BKK && (KPT == 2 || KPT == 4) && !MF16), "quad loads: tap-major 32x32x16 tiles");
    static_assert(!F32 || (!MF16 && !TAPK && !B4 && !PSB), "fp32 operands: table path, 32x32 tiles");
    constexpr int LDKF = BK + 4;     // fp32 rows: 144 bytes (16-byte aligned; 32 lanes x 16 bytes cover every bank 4 times)
    __shared__ __attribute__((aligned(16))) unsigned short Ahi[2][F32 ? 8 : BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Alo[2][F32 ? 8 : BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Bhi[2][F32 ? 8 : BROWS * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Blo[2][F32 ? 8 : BROWS * LDK];
    __shared__ __attribute__((aligned(16))) float Af[2][F32 ? BM * LDKF : 4];
    __shared__ __attribute__((aligned(16))) float Bf[2][F32 ? BN * LDKF : 4];
    __shared__ int2 tab[TAPK ? 1 : 4][BK];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPW;
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = p.R * p.S;
    const int SHW = p.H * p.W;

    const int nk_all = (p.KG + BK - 1) / BK;
    const int kt0 = (p.splitk > 1) ? blockIdx.y * p.kchunk : 0;
    const int nk = (p.splitk > 1) ? min(nk_all, kt0 + p.kchunk) : nk_all;
    if (kt0 >= nk) return;  // uniform per workgroup

    // ------------------------------------------------------------------ pipeline
    // The two roles are separate code regions (register live ranges do not add up); each executes exactly the same
    // sequence of workgroup barriers: 2 in the prologue, then one per K-tile.
    if (producer) {
        // ------------------------------------------------------------------ producer state (threads 0..255)
        const __amdgpu_buffer_rsrc_t ars = make_rsrc(p.a_src, p.a_bytes), brs = make_rsrc(p.b_src, p.b_bytes);
        const int ptid = tid & (NP - 1);
        // A: thread -> ACOLS columns (m), one k part of AK consecutive k each
        const int acol0 = ptid % (BM / ACOLS), apart = ptid / (BM / ACOLS) % AP;
        // B: thread -> BCOLS pixel columns, one k part of BKK consecutive k each
        const int bcol0 = ptid % (BN / BCOLS), bpart = ptid / (BN / BCOLS) % BP;
        int bbase[BCOLS];
        unsigned vmask[BCOLS];
#pragma unroll
        for (int c = 0; c < BCOLS; ++c) {
            const int pix = n0 + bcol0 + c * (BN / BCOLS);
            vmask[c] = 0;
            bbase[c] = 0;
            if (pix < p.NG) {
                const int n_img = pix / (p.OH * p.OW);
                const int prem = pix - n_img * (p.OH * p.OW);
                const int py = prem / p.OW, px = prem - py * p.OW;
                const int y0 = py * p.stride - p.pad, x0 = px * p.stride - p.pad_w;
                for (int r = 0; r < p.R; ++r)
                    for (int s2 = 0; s2 < p.S; ++s2)
                        if ((unsigned)(y0 + r) < (unsigned)p.H && (unsigned)(x0 + s2) < (unsigned)p.W) vmask[c] |= 1u << (r * p.S + s2);
                bbase[c] = n_img * p.C * SHW + (y0 * p.W + x0);
            }
        }
        // B4: quad state (pixels 4 quad .. 4 quad + 3 of the tile are in one output row; stride 1)
        const int quad = ptid % QN, kpart = ptid / QN;      // a wave = the 64 quads of one channel part
        int qbase = 0, qx0 = 0;
        unsigned qrmask = 0, qemask = 0;       // qemask: nibble s = the quad elements whose column x0 + s + j is outside [0, W)
        if constexpr (B4) {
            const int pix = n0 + 4 * quad;
            if (pix < p.NG) {
                const int n_img = pix / (p.OH * p.OW);
                const int prem = pix - n_img * (p.OH * p.OW);
                const int py = prem / p.OW, px = prem - py * p.OW;
                const int y0 = py - p.pad;
                qx0 = px - p.pad_w;
                for (int r = 0; r < p.R; ++r)
                    if ((unsigned)(y0 + r) < (unsigned)p.H) qrmask |= 1u << r;
                qbase = (n_img * p.C + kpart * KPT) * SHW + y0 * p.W + qx0;
                for (int t = 0; t < p.S; ++t)
                    for (int j = 0; j < 4; ++j)
                        if ((unsigned)(qx0 + t + j) >= (unsigned)p.W) qemask |= 1u << (4 * t + j);
            }
        }
        auto build_tab = [&](int kt) {
            if (tid < BK) {
                const int kg = kt * BK + tid;
                int c, r, s2;
                decode_crs(kg, RS, p.S, c, r, s2);
                const unsigned bit = (kg < p.KG) ? (1u << (r * p.S + s2)) : 0u;
                tab[kt & 3][tid] = make_int2(c * SHW + r * p.W + s2, (int)bit);
            }
        };
        auto fetch = [&](int kt, unsigned (&ra)[ACOLS][AK], float (&rb)[BCOLS][BKK]) {
            // all table entries of this thread's k part first, in one batch: left to itself the compiler reads each entry
            // inside the address select of its load (two dependent LDS round trips and an exec branch per load; 173 cycles
            // per B load against 44 per A load on the in-kernel clocks of tools/ubench/igemm_stamps.hip)
            int2 e[BKK];
#pragma unroll
            for (int i = 0; i < BKK; ++i) e[i] = tab[kt & 3][BKK * bpart + i];
            asm volatile("" ::: "memory");
            const int ka = kt * BK + AK * apart;
#pragma unroll
            for (int c = 0; c < ACOLS; ++c) {
                const int m = m0 + acol0 + c * (BM / ACOLS);
#pragma unroll
                for (int i = 0; i < AK; ++i) ra[c][i] = bload_u32(ars, (m < p.M) ? (unsigned)(m + (ka + i) * p.M) * 4u : 0xFFFFFFFFu);
            }
#pragma unroll
            for (int i = 0; i < BKK; ++i) {
#pragma unroll
                for (int c = 0; c < BCOLS; ++c) {
                    const unsigned off = (unsigned)(bbase[c] + e[i].x) * 4u;
                    rb[c][i] = bload_f32(brs, (vmask[c] & (unsigned)e[i].y) ? off : 0xFFFFFFFFu);
                }
            }
        };
        // TAPK: the reduction index is ordered by tap_major_row() (the launcher lays the weights out to match) and C % BK == 0,
        // so the BK k's of a K-tile share one filter tap and differ in the channel only.  Per K-tile a thread then needs ONE
        // voffset per operand (A: its row; B: its pixel moved by the tap, or out of range when the tap falls into the padding)
        // and every load adds a wave-uniform SGPR offset (k * M resp. channel * H * W) in the instruction's soffset field: no
        // table and no per-load address arithmetic on the vector ALU, whose issue port the consumers' MFMAs share (16x16x32:
        // 8 of every 16 cycles); ~40 VALU instructions per K-tile and thread instead of ~110.  The hardware range-checks
        // voffset + soffset (tools/ubench/soffset_probe.hip), so K-tiles past the end of the reduction (the pipeline runs up to
        // three ahead) read zeros like every masked element; `live` only keeps their offsets from wrapping.
        constexpr unsigned kOut = 0xFFFFFFF0u;
        int f_kt = kt0, f_c = 0, f_r = 0, f_s = 0;       // scalar state: the next K-tile to fetch, its first channel and its tap
        if constexpr (TAPK) {       // K-tile index = (channel chunk * R + r) * S + s  (tap_major_row)
            const int q = kt0 / p.S, ch = q / p.R;
            f_s = kt0 - q * p.S;
            f_r = q - ch * p.R;
            f_c = ch * BK;
        }
        const int apart_s = (wave * 64 / (BM / ACOLS)) % AP, bpart_s = (wave * 64 / (BN / BCOLS)) % BP;   // = apart, bpart (wave-uniform)
        auto fetch_tap = [&](unsigned (&ra)[ACOLS][AK], float (&rb)[BCOLS][BKK], unsigned& emask) {
            const bool live = f_kt < nk_all;
            const unsigned c4 = (unsigned)SHW * 4u;
            // A: weights as two bf16 planes, hi then lo, each [KG/8][M][8] (eight consecutive k of a row are 16 contiguous bytes;
            // pack_fwd_tap_kernel / pack_dgrad_tap_kernel): one 16-byte load per plane and 8 k, stored to LDS as loaded (no
            // v_perm: producer VALU beside the prioritised MFMAs costs ~1 % of the 128 x 256 tile's time per instruction).
            // The CU's texture addresser takes ~10 cycles per dword wave-load and ~20 per dwordx4 one; with dword loads the
            // second producer wave of each SIMD spent 2 000 of a K-tile's 3 000 cycles in the load-issue phase (per-wave
            // clocks, tools/ubench/igemm_stamps.hip PER_WAVE=1).
            const unsigned m16 = (unsigned)p.M * 16u;
            const unsigned sa = (unsigned)((f_kt * BK + AK * apart_s) >> 3) * m16;
            const unsigned lo_plane = (unsigned)(p.KG >> 3) * m16;
#pragma unroll
            for (int c = 0; c < ACOLS; ++c) {
                const int m = m0 + acol0 + c * (BM / ACOLS);
                const unsigned vo = (live & (m < p.M)) ? (unsigned)m * 16u : kOut;
#pragma unroll
                for (int i = 0; i < AK / 8; ++i) {
                    const u32x4 vh = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, sa + i * m16, 0);
                    const u32x4 vl = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, sa + i * m16 + lo_plane, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ra[c][8 * i + e] = vh[e]; ra[c][8 * i + 4 + e] = vl[e]; }
                }
            }

            if constexpr (B4) {
                // rb[0][4 i + j] = channel (f_c + kpart KPT + i), pixel j of the quad.  Elements whose tap column falls outside
                // [0, W) are cleared in stash (emask); a tap row outside [0, H) invalidates the whole quad.
                const bool rowok = live & (((qrmask >> f_r) & 1u) != 0u);
                const int o = qbase + f_r * p.W + f_s;            // element offset without the chunk term (soffset)
                const unsigned sb = (unsigned)f_c * c4;
                emask = (qemask >> (4 * f_s)) & 15u;
                // The quad of a launch that starts before the tensor's first element (image 0, channel part 0, row 0, x < 0) cannot
                // use its negative voffset (out of range whatever soffset adds): it loads from element 0 and stash moves the
                // registers up by -o places (emask bits 4-5).  No branch around the loads: that made the compiler wait for
                // vmcnt(0..3) in this loop, i.e. for the loads issued half an iteration earlier.
                int oc = o;
                if (__builtin_expect(rowok & (o < 0), 0)) { emask |= (unsigned)((-o) & 3) << 4; oc = 0; }
                const unsigned vo = rowok ? (unsigned)oc * 4u : kOut;
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(brs, vo, sb + i * c4, 0);
                    rb[0][4 * i] = __uint_as_float(v[0]); rb[0][4 * i + 1] = __uint_as_float(v[1]);
                    rb[0][4 * i + 2] = __uint_as_float(v[2]); rb[0][4 * i + 3] = __uint_as_float(v[3]);
                }
            } else {
                const unsigned sb = (unsigned)(f_c + BKK * bpart_s) * c4;
                const int toff = f_r * p.W + f_s, tbit = f_r * p.S + f_s;
#pragma unroll
                for (int c = 0; c < BCOLS; ++c) {
                    const unsigned vo = (live & (((vmask[c] >> tbit) & 1u) != 0u)) ? (unsigned)(bbase[c] + toff) * 4u : kOut;
#pragma unroll
                    for (int i = 0; i < BKK; ++i)
                        rb[c][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, vo, sb + i * c4, 0));
                }
            }
            ++f_kt;
            if (++f_s == p.S) {
                f_s = 0;
                if (++f_r == p.R) { f_r = 0; f_c += BK; }
            }
        };
        auto stash = [&](int buf, const unsigned (&ra)[ACOLS][AK], float (&rb)[BCOLS][BKK], unsigned emask) {
            if constexpr (F32) {        // fp32 words as loaded
#pragma unroll
                for (int c = 0; c < ACOLS; ++c)
#pragma unroll
                    for (int g = 0; g < AK / 4; ++g)
                        *reinterpret_cast<uint4*>(&Af[buf][(acol0 + c * (BM / ACOLS)) * LDKF + AK * apart + 4 * g]) =
                            make_uint4(ra[c][4 * g], ra[c][4 * g + 1], ra[c][4 * g + 2], ra[c][4 * g + 3]);
#pragma unroll
                for (int c = 0; c < BCOLS; ++c)
#pragma unroll
                    for (int g = 0; g < BKK / 4; ++g)
                        *reinterpret_cast<float4*>(&Bf[buf][(bcol0 + c * (BN / BCOLS)) * LDKF + BKK * bpart + 4 * g]) =
                            make_float4(rb[c][4 * g], rb[c][4 * g + 1], rb[c][4 * g + 2], rb[c][4 * g + 3]);
                return;
            }
#pragma unroll
            for (int c = 0; c < ACOLS; ++c)
#pragma unroll
                for (int g = 0; g < AK / 8; ++g) {
                    unsigned h[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (TAPK) {       // hi / lo planes: the words are LDS images already
                            h[i] = ra[c][8 * g + i];
                            l[i] = ra[c][8 * g + 4 + i];
                        } else {
                            h[i] = __builtin_amdgcn_perm(ra[c][8 * g + 2 * i + 1], ra[c][8 * g + 2 * i], 0x07060302u);
                            l[i] = __builtin_amdgcn_perm(ra[c][8 * g + 2 * i + 1], ra[c][8 * g + 2 * i], 0x05040100u);
                        }
                    }
                    const int off = (acol0 + c * (BM / ACOLS)) * LDK + AK * apart + 8 * g;
                    *reinterpret_cast<uint4*>(&Ahi[buf][off]) = make_uint4(h[0], h[1], h[2], h[3]);
                    *reinterpret_cast<uint4*>(&Alo[buf][off]) = make_uint4(l[0], l[1], l[2], l[3]);
                }
            if constexpr (B4) {
                if (emask) {     // border quads only
                    const unsigned sh = emask >> 4;
                    if (sh) {
#pragma unroll
                        for (int i = 0; i < KPT; ++i)
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                if (sh > (unsigned)q) {
                                    rb[0][4 * i + 3] = rb[0][4 * i + 2]; rb[0][4 * i + 2] = rb[0][4 * i + 1];
                                    rb[0][4 * i + 1] = rb[0][4 * i]; rb[0][4 * i] = 0.f;
                                }
                    }
#pragma unroll
                    for (int i = 0; i < KPT; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((emask >> j) & 1u) rb[0][4 * i + j] = 0.f;
                }
                // 8-byte stores (4 channels of one pixel).  Exchanging pixel halves between lanes l / l + 32 (v_permlane32_swap) to
                // keep 16-byte stores removes the 2-way bank conflicts of these (22 % of LDS cycles -> 0) but its 8 VALU
                // instructions cost more (-2 % against dword loads; producer VALU beside prioritised MFMAs ~ 1 % each).
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = (j * GR + quad) * LDK + kpart * KPT;
                    if constexpr (KPT == 4) {
                        unsigned h0, l0, h1, l1;
                        split_or_regroup<PSB, H16>(rb[0][j], rb[0][4 + j], h0, l0);
                        split_or_regroup<PSB, H16>(rb[0][8 + j], rb[0][12 + j], h1, l1);
                        *reinterpret_cast<uint2*>(&Bhi[buf][off]) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(&Blo[buf][off]) = make_uint2(l0, l1);
                    } else {      // 256 x 128 tile: 2 channels per thread
                        unsigned h0, l0;
                        split_or_regroup<PSB, H16>(rb[0][j], rb[0][4 + j], h0, l0);
                        *reinterpret_cast<unsigned*>(&Bhi[buf][off]) = h0;
                        *reinterpret_cast<unsigned*>(&Blo[buf][off]) = l0;
                    }
                }
            } else {
#pragma unroll
            for (int c = 0; c < BCOLS; ++c)
#pragma unroll
                for (int g = 0; g < BKK / 8; ++g) {
                    unsigned h[4], l[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        split_or_regroup<PSB, H16>(rb[c][8 * g + 2 * i], rb[c][8 * g + 2 * i + 1], h[i], l[i]);
                    }
                    const int off = (bcol0 + c * (BN / BCOLS)) * LDK + BKK * bpart + 8 * g;
                    *reinterpret_cast<uint4*>(&Bhi[buf][off]) = make_uint4(h[0], h[1], h[2], h[3]);
                    *reinterpret_cast<uint4*>(&Blo[buf][off]) = make_uint4(l[0], l[1], l[2], l[3]);
                }
            }
        };

        unsigned ra0[ACOLS][AK], ra1[ACOLS][AK];
        float rb0[BCOLS][BKK], rb1[BCOLS][BKK];
        // Register set 0 / 1 holds the even / odd K-tiles (relative to kt0).  Iteration t splits + stores tile t (loaded two
        // iterations ago) into LDS buffer t&1, then refills the same registers with tile t+2 and meets the consumers at the
        // barrier; the consumers multiply tile t after that barrier.  Every load has two full iterations to land and a
        // register set is never written while its loads are in flight.  (The loop starts on the older set and the two
        // prologue fetches are pinned in order, so that the compiler's vmcnt waits at the loop head stay partial.)
        if constexpr (!TAPK) {
            build_tab(kt0);
            build_tab(kt0 + 1);
            build_tab(kt0 + 2);
            build_tab(kt0 + 3);
        }
        __syncthreads();
        unsigned em0 = 0, em1 = 0;       // B4: border-element masks of the two register sets
        if constexpr (TAPK) fetch_tap(ra0, rb0, em0); else fetch(kt0, ra0, rb0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAPK) fetch_tap(ra1, rb1, em1); else fetch(kt0 + 1, ra1, rb1);
        __builtin_amdgcn_sched_barrier(0);
#ifdef AG2V_STAMPS
        unsigned long long sa[3] = {0, 0, 0};
#define STAMP(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(v)
#endif
        // No exit in the middle of the pair: a K-range with an odd tile count runs one surplus half-iteration (it stores a
        // tile the consumers never multiply).  A mid-loop break merges "set 0 just refilled" into the loop-head state and the
        // compiler then drains vmcnt to 0 at the top of every iteration.
        for (int kt = kt0; kt < nk; kt += 2) {
            STAMP(c0)
#ifndef AG2V_DIAG_NO_PRODUCER_WORK      // tools/ubench/igemm_stamps.hip: consumers alone / producers alone
            stash(0, ra0, rb0, em0);
#endif
            STAMP(c1)
#ifndef AG2V_DIAG_NO_PRODUCER_WORK
            if constexpr (TAPK) fetch_tap(ra0, rb0, em0);
            else { build_tab(kt + 4); fetch(kt + 2, ra0, rb0); }
#endif
            STAMP(c2)
            __syncthreads();
            STAMP(c3)
#ifdef AG2V_STAMPS
            sa[0] += c2 - c1; sa[1] += c1 - c0; sa[2] += c3 - c2;
#endif
            STAMP(d0)
#ifndef AG2V_DIAG_NO_PRODUCER_WORK
            stash(1, ra1, rb1, em1);
#endif
            STAMP(d1)
#ifndef AG2V_DIAG_NO_PRODUCER_WORK
            if constexpr (TAPK) fetch_tap(ra1, rb1, em1);
            else { build_tab(kt + 5); fetch(kt + 3, ra1, rb1); }
#endif
            STAMP(d2)
            __syncthreads();
            STAMP(d3)
#ifdef AG2V_STAMPS
            sa[0] += d2 - d1; sa[1] += d1 - d0; sa[2] += d3 - d2;
#endif
        }
#ifdef AG2V_STAMPS
        if (lane == 0 && p.stamps) {
            unsigned long long* o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 4;
            o[0] = sa[0]; o[1] = sa[1]; o[2] = sa[2]; o[3] = nk - kt0;
        }
#endif
        return;
    }

    // ------------------------------------------------------------------ consumer state (waves 4..7)
    const int cw = wave - NPW;
    // The vector-issue port of a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md, two waves per SIMD); the
    // producers are the older waves, so at equal priority their VALU / address work goes first and the MFMAs wait.  With
    // the lean TAPK producers (16-byte A loads, no address VALU) the MFMA waves take priority 2: +4..13 % per layer.  (With
    // the table-path producers the same priority starves the loads: -10 %.)
    if constexpr (TAPK) __builtin_amdgcn_s_setprio(2);
    const int wm0 = (cw / CWN) * (TM * 32), wn0 = (cw % CWN) * (TN * 32);
    // Two MFMA shapes.  32x32x16 (two k-steps per K-tile) and 16x16x32 (one instruction spans the K-tile; lane l holds row
    // l & 15, k = 8 (l >> 4) .. + 7; C/D: col = lane & 15, row = 4 (lane >> 4) + reg).  Same LDS reads and MFMA cycles per
    // K-tile; the chip holds a higher clock on the 16x16 shape but it occupies the SIMD's issue port for 8 of its 16 cycles:
    // measured +3..6 % on the 256x128 tile and -16 % on the producer-heavier 128x256 tile, so MF16 is set for the former only.
    constexpr int EM = MF16 ? TM * 2 : TM, EN = MF16 ? TN * 2 : TN;     // accumulator tiles per consumer wave
    constexpr int ER = MF16 ? 16 : 32;                                  // rows = columns of one accumulator tile
    constexpr int EREGS = MF16 ? 4 : 16;
    typedef float accv __attribute__((ext_vector_type(EREGS)));
    accv acc[EM][EN];
#pragma unroll
    for (int i = 0; i < EM; ++i)
#pragma unroll
        for (int j = 0; j < EN; ++j)
#pragma unroll
            for (int r = 0; r < EREGS; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & (ER - 1), fq = lane / ER;
    auto compute = [&](int buf) {
        if constexpr (F32) {
            const int half = lane >> 5, r32 = lane & 31;
            float4 bfr[EN][4];
#pragma unroll
            for (int j = 0; j < EN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bfr[j][q] = *reinterpret_cast<const float4*>(&Bf[buf][(wn0 + j * 32 + r32) * LDKF + 16 * half + 4 * q]);
#pragma unroll
            for (int i = 0; i < EM; ++i) {
                float4 afr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    afr[q] = *reinterpret_cast<const float4*>(&Af[buf][(wm0 + i * 32 + r32) * LDKF + 16 * half + 4 * q]);
#pragma unroll
                for (int j = 0; j < EN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].x, bfr[j][q].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].y, bfr[j][q].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].z, bfr[j][q].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].w, bfr[j][q].w, acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < (MF16 ? 1 : BK / 16); ++ks) {
            bf16x8 bhi[EN], blo[EN];
#pragma unroll
            for (int j = 0; j < EN; ++j) {
                const int pix = wn0 + j * ER + fr;
                const int o = (B4 ? (pix & 3) * GR + (pix >> 2) : pix) * LDK + ks * 16 + 8 * fq;
                bhi[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                blo[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
#pragma unroll
            for (int i = 0; i < EM; ++i) {
                const int o = (wm0 + i * ER + fr) * LDK + ks * 16 + 8 * fq;
                const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
                const bf16x8 alo = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
#pragma unroll
                for (int j = 0; j < EN; ++j) {
                    if constexpr (MF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bhi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, blo[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bhi[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = mfma3<H16>(ahi, alo, bhi[j], blo[j], acc[i][j]);
                    }
                }
            }
        }
    };

    __syncthreads();
#ifdef AG2V_STAMPS
    unsigned long long sc[2] = {0, 0};
#endif
    const int nk2 = kt0 + ((nk - kt0 + 1) & ~1);   // the producers run an even number of iterations
    // SKEW (32x32x16 tiles): the last MFMAs of a K-tile -- row block 1 of its second k-step -- are issued AFTER the next barrier,
    // from fragments read before it, right behind the first LDS reads of the next tile: all eight consumer waves leave the
    // barrier together and used to wait ~450 of a K-tile's ~2 700 cycles for those reads with an idle MFMA pipe.  Every
    // accumulator still receives its products in the same order (bit-identical results).  c3 step: 822.7 -> 816.6 ms (two builds,
    // same box).  Requesting the second row block's fragments before the deferred MFMAs as well (128 VGPRs) was slower again, and
    // the fp32 form (64 MFMAs per K-tile) gains nothing from a deferred quarter.
    constexpr bool SKEW = !MF16 && !F32 && (BK / 16 == 2) && EM == 2 && (AG2V_CONSUMER_SKEW != 0);
    if constexpr (SKEW) {
        auto ldB = [&](int buf, int ks, bf16x8 (&bh)[EN], bf16x8 (&bl)[EN]) {
#pragma unroll
            for (int j = 0; j < EN; ++j) {
                const int pix = wn0 + j * ER + fr;
                const int o = (B4 ? (pix & 3) * GR + (pix >> 2) : pix) * LDK + ks * 16 + 8 * fq;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
        };
        auto ldA = [&](int buf, int ks, int i, bf16x8& ah, bf16x8& al) {
            const int o = (wm0 + i * ER + fr) * LDK + ks * 16 + 8 * fq;
            ah = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
            al = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
        };
        auto mm = [&](int i, const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[EN], const bf16x8 (&bl)[EN]) {
#pragma unroll
            for (int j = 0; j < EN; ++j) acc[i][j] = mfma3<H16>(ah, al, bh[j], bl[j], acc[i][j]);
        };
        bf16x8 hbh[EN], hbl[EN], hah, hal;      // deferred: (row block 1, k-step 1) of the previous tile
        bool have = false;
        for (int kt = kt0; kt < nk2; ++kt) {
            __syncthreads();   // tile kt is in LDS buffer (kt - kt0) & 1
            if (kt < nk) {
                const int buf = (kt - kt0) & 1;
                bf16x8 bh0[EN], bl0[EN], a0h, a0l, a1h, a1l;
                ldB(buf, 0, bh0, bl0);
                ldA(buf, 0, 0, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                if (have) mm(1, hah, hal, hbh, hbl);          // previous tile's tail, while the reads above are in flight
                __builtin_amdgcn_sched_barrier(0);
                ldA(buf, 0, 1, a1h, a1l);
                mm(0, a0h, a0l, bh0, bl0);
                ldB(buf, 1, hbh, hbl);
                ldA(buf, 1, 0, a0h, a0l);
                mm(1, a1h, a1l, bh0, bl0);
                ldA(buf, 1, 1, hah, hal);
                mm(0, a0h, a0l, hbh, hbl);
                have = true;
            }
        }
        if (have) mm(1, hah, hal, hbh, hbl);
    } else
    for (int kt = kt0; kt < nk2; ++kt) {
        STAMP(e0)
        __syncthreads();   // tile kt is in LDS buffer (kt - kt0) & 1
        STAMP(e1)
#ifndef AG2V_DIAG_NO_CONSUMER_WORK
        if (kt < nk) compute((kt - kt0) & 1);
#endif
        STAMP(e2)
#ifdef AG2V_STAMPS
        sc[0] += e2 - e1; sc[1] += e1 - e0;
#endif
    }
#ifdef AG2V_STAMPS
    if (lane == 0 && p.stamps) {
        unsigned long long* o = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 4;
        o[0] = sc[0]; o[1] = sc[1]; o[2] = 0; o[3] = nk - kt0;
    }
#endif
#undef STAMP

    // ---- epilogue (consumers).  The accumulator tiles put a lane's pixel in (lane & 15) or (lane & 31) and spread a row's 64
    // pixels over 4 (or 2) registers: stored as they are, every store instruction writes 4 x 64-byte (2 x 128-byte) pieces of
    // different rows, and the in-kernel clocks show 20 000 cycles per tile in this epilogue (every CU's 128 KB tile reaches
    // HBM in such pieces at the same moment).  A 4x4 (2x2) transpose of 16-lane (32-lane) blocks between those registers --
    // v_permlane32_swap + v_permlane16_swap (tools/ubench/permlane_probe.hip) -- gives every register one row and 64
    // consecutive pixels, i.e. one 256-byte run per store instruction.
    static_assert(EN * ER == 64, "a consumer wave owns 64 pixels");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int gplane = p.OH * p.OW;
    const int plane = p.o_H * p.o_W;
    const bool mapped = (p.o_mul != 1 || p.o_add_y != 0 || p.o_add_x != 0 || p.o_W != p.OW);
    const int opix = n0 + wn0 + lane;
    bool pvalid = opix < p.NG;
    const int on = pvalid ? opix / gplane : 0;
    int orem = pvalid ? opix - on * gplane : 0;
    if (mapped) {
        const int qy = orem / p.OW, qx = orem - qy * p.OW;
        const int yy = qy * p.o_mul + p.o_add_y, xx = qx * p.o_mul + p.o_add_x;
        pvalid = pvalid && (unsigned)yy < (unsigned)p.o_H && (unsigned)xx < (unsigned)p.o_W;
        orem = yy * p.o_W + xx;
    }
    float* obase = p.out + (size_t)on * p.out_nstride + orem;
    const float* rbase = p.residual ? p.residual + (size_t)on * p.M * plane + orem : nullptr;
    const float* mbase = (p.mask && p.splitk == 1) ? p.mask + (size_t)on * p.M * plane + orem : nullptr;
    const float gs = (p.gscale ? p.gscale[on / p.group_n] : 1.f) * (p.oscale != 0.f ? p.oscale : 1.f);   // 1/sigma_t (x 1/weight scale of the h16 pack)
    auto emit = [&](int m, float v) {
        if (pvalid && m < p.M) {
            v *= gs;
            if (kt0 == 0) {
                if (p.bias) v += p.bias[m];
                if (rbase) v += rbase[(size_t)m * plane];
            }
            if (p.splitk == 1) {   // (a split-K launch applies these in a pass over the finished sums: launch_act_mask)
                if (p.relu) v = fmaxf(v, 0.f);
                if (mbase) v = (mbase[(size_t)m * plane] > 0.f) ? v : 0.f;
                if (p.out_split) v = pack_split(v);
            }
            if (p.splitk > 1)
                atomicAdd(obase + (size_t)m * plane, v);
            else
                obase[(size_t)m * plane] = v;
        }
    };
#pragma unroll
    for (int i = 0; i < EM; ++i) {
#pragma unroll
        for (int r = 0; r < (MF16 ? EREGS : 0); ++r) {
            if constexpr (MF16) {
                unsigned a0 = __float_as_uint(acc[i][0][r]), a1 = __float_as_uint(acc[i][1][r]);
                unsigned a2 = __float_as_uint(acc[i][2][r]), a3 = __float_as_uint(acc[i][3][r]);
                u32x2 t;
                t = __builtin_amdgcn_permlane32_swap(a0, a2, false, false); a0 = t[0]; a2 = t[1];
                t = __builtin_amdgcn_permlane32_swap(a1, a3, false, false); a1 = t[0]; a3 = t[1];
                t = __builtin_amdgcn_permlane16_swap(a0, a1, false, false); a0 = t[0]; a1 = t[1];
                t = __builtin_amdgcn_permlane16_swap(a2, a3, false, false); a2 = t[0]; a3 = t[1];
                const int mrow = m0 + wm0 + i * 16 + r;      // register g now holds row mrow + 4 g, pixel wn0 + lane
                emit(mrow, __uint_as_float(a0));
                emit(mrow + 4, __uint_as_float(a1));
                emit(mrow + 8, __uint_as_float(a2));
                emit(mrow + 12, __uint_as_float(a3));
            }
        }
        if constexpr (!MF16) {
            // 32x32x16 tiles: registers r = 4 g .. 4 g + 3 hold rows 8 g .. 8 g + 7 of this 32-row block after the swap.  The
            // epilogue's reads are batched: one vector load brings the block's 32 bias values (a row's value is then a
            // v_readlane), and the residual / mask words of 8 rows are requested together before the first of them is used.
            // (Row by row, each of the up to three conditional loads was followed by s_waitcnt vmcnt(0): 64-190 serialised
            // memory round trips per wave and tile.)
            const int mblk = m0 + wm0 + i * 32;
            float bvec = 0.f;
            if (kt0 == 0 && p.bias) {
                const int mb = mblk + (lane & 31);
                bvec = (mb < p.M) ? p.bias[mb] : 0.f;
            }
            const float* aux = rbase ? rbase : mbase;       // at most one of them (ag2v_conv2d_dgrad_acc rejects both)
            const float ninf = -__builtin_huge_valf();
            const float rmul = (rbase && kt0 == 0) ? 1.f : 0.f;                         // residual added by the first K chunk only
            const float relu_floor = (p.relu && p.splitk == 1) ? 0.f : ninf;
            const float mask_thr = mbase ? 0.f : ninf;                                  // no mask: every word passes
            const bool osplit = p.out_split && p.splitk == 1;                           // (split-K: launch_act_mask packs)
            // the residual / mask words of all 32 rows are requested before the first store of the block goes out (a wait for
            // loads issued after stores would wait for those stores as well: vmcnt counts in issue order)
            float axa[4][8];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 8; ++k) axa[g][k] = 0.f;
            if (aux) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int m = mblk + 8 * g + k;
                        if (pvalid && m < p.M) axa[g][k] = aux[(size_t)m * plane];
                    }
                __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) once per 32-row block
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[8];
                const float (&ax)[8] = axa[g];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned a0 = __float_as_uint(acc[i][0][4 * g + q]), a1 = __float_as_uint(acc[i][1][4 * g + q]);
                    const u32x2 t = __builtin_amdgcn_permlane32_swap(a0, a1, false, false);
                    v[q] = __uint_as_float(t[0]);            // row mblk + 8 g + q
                    v[q + 4] = __uint_as_float(t[1]);        // row mblk + 8 g + q + 4
                }
                // straight-line arithmetic on the loaded words (uniform multipliers / thresholds instead of branches: a use of a
                // loaded register inside a conditional block makes the compiler wait for vmcnt(0) there, stores included)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float x = fmaf(v[k], gs, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bvec), 8 * g + k)));
                    x = fmaf(ax[k], rmul, x);
                    x = fmaxf(x, relu_floor);
                    v[k] = (ax[k] > mask_thr) ? x : 0.f;
                }
                if (osplit) {       // (wave-uniform) the activation goes to bf16x3 convolutions only: stored as split words
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = pack_split(v[k]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = mblk + 8 * g + k;
                    if (pvalid && m < p.M) {
                        if (p.splitk > 1)
                            atomicAdd(obase + (size_t)m * plane, v[k]);
                        else
                            obase[(size_t)m * plane] = v[k];
                    }
                }
            }
        }
    }
}
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16, bool TAPK, bool B4>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_igemm_bf16x3_ws_kernel(ConvArgs p) {
    igemm_bf16x3_body<BM, BN, CWM, CWN, NPW, MF16, TAPK, B4, false>(p);
}
// exact fp32 on the same pipeline (ConvArgs::f32): conv_igemm_kernel's arithmetic at a higher MFMA occupancy
template <int BM, int BN, int CWM, int CWN, int NPW>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_igemm_f32_ws_kernel(ConvArgs p) {
    igemm_bf16x3_body<BM, BN, CWM, CWN, NPW, false, false, false, false, true>(p);
}
// fp16 3-term split (ConvArgs::h16): forward passes on normalised operands at fp32-class accuracy
template <int BM, int BN, int CWM, int CWN, int NPW, bool TAPK, bool B4>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_igemm_f16x3_ws_kernel(ConvArgs p) {
    igemm_bf16x3_body<BM, BN, CWM, CWN, NPW, false, TAPK, B4, false, false, true>(p);
}
// the same tiles reading a split-word gathered tensor (ConvArgs::b_split)
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16, bool TAPK, bool B4>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_igemm_bf16x3_ps_kernel(ConvArgs p) {
    igemm_bf16x3_body<BM, BN, CWM, CWN, NPW, MF16, TAPK, B4, true>(p);
}

// Weight gradient: dW[m,j] = sum_q dY[m,q] * Xg[q,j], q = (n,oy,ox), j = (c,r,s).  Both operands are activations (split
// in-kernel).  General form (any stride / width): a producer thread owns one PAIR of consecutive pixels q for a set of rows of dY
// and columns of the gather, so each (hi | lo) pair is one packed ds_write_b32 into the k-contiguous LDS rows; 8 consumer
// waves (64x64 each) only read fragments and issue MFMAs (same producer / consumer split as conv_igemm_bf16x3_ws_kernel).
template <int BM, int BN, int CWM, int CWN, int NPW>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_wgrad_bf16x3_ws_kernel(ConvArgs p) {
    constexpr int BK = 32, LDK = 40, NP = NPW * 64;
    constexpr int TM = BM / CWM / 32, TN = BN / CWN / 32;
    constexpr int NRG = NP / 16;                     // row groups: a producer thread = (pixel pair, row group)
    constexpr int RA = BM / NRG, RB = BN / NRG;      // rows of dY / gather columns per producer thread
    static_assert(BM % NRG == 0 && BN % NRG == 0, "producer mapping");
    __shared__ __attribute__((aligned(16))) unsigned short Ahi[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Alo[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Bhi[2][BN * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Blo[2][BN * LDK];
    __shared__ int4 qtab[4][BK];   // per pixel q of a K-tile: {x offset of tap (0,0), dY offset, (yb<<16)|xb, valid}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPW;
    // 1-D grid over (pixel-range split, tile), consecutive ids on one XCD: the tiles of one split read the same dY rows (and
    // neighbouring x channels), so they share that XCD's L2 instead of fetching dY once per tile (PMC: 4.8 GB -> see profiles)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int gl = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int split = gl / ntiles;
    const int lid = gl - split * ntiles;
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = p.R * p.S;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    // pixel range of this workgroup: split sl of sample group grp (groups == 1: the whole batch is one group)
    const int grp = split / p.spg, sl = split - grp * p.spg;
    const int qs = grp * p.gq + sl * p.kchunk;
    const int qe = min(qs + p.kchunk, (grp + 1) * p.gq);
    if (qs >= qe) return;
    const int nt = (qe - qs + BK - 1) / BK;
    float* const outg = p.det ? p.out + (size_t)split * ((size_t)p.M * p.NG) : p.out + (size_t)grp * p.out_gstride;
    const bool atomic_out = !p.det && (p.spg > 1 || p.accumulate);

    if (producer) {
        const __amdgpu_buffer_rsrc_t ars = make_rsrc(p.a_src, p.a_bytes), brs = make_rsrc(p.b_src, p.b_bytes);
        const int qp = tid & 15, rg = tid >> 4;   // pixel pair (2qp, 2qp+1) of the K-tile; rows / columns rg + NRG i
        int aoff[RA], boff[RB], bpk[RB];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + rg + NRG * i;
            aoff[i] = (m < p.M) ? m * OHW : -1;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int j = n0 + rg + NRG * i;
            int c, r, s;
            decode_crs(j, RS, p.S, c, r, s);
            bpk[i] = ((r - p.pad + 64) & 255) | (((s - p.pad + 64) & 255) << 8) | ((j < p.NG) ? (1 << 16) : 0);
            boff[i] = c * HW + (r - p.pad) * p.W + (s - p.pad);
        }
        auto build_tab = [&](int t) {   // t = tile index relative to qs
            if (tid < BK) {
                const int q = qs + t * BK + tid;
                const bool v = q < qe;
                const int qc = v ? q : qs;
                const int n = qc / OHW;
                const int pixl = qc - n * OHW;
                const int oy = pixl / p.OW, ox = pixl - oy * p.OW;
                const int yb = oy * p.stride, xb = ox * p.stride;
                qtab[t & 3][tid] = make_int4(n * p.C * HW + yb * p.W + xb, n * p.K * OHW + pixl, (yb << 16) | xb, v ? 1 : 0);
            }
        };
        auto fetch = [&](int t, float (&ra)[2 * RA], float (&rb)[2 * RB]) {
            // both table entries first; validity as bitwise (branch-free) predicates: with && the compiler builds an exec
            // branch around every load
            const int4 e0 = qtab[t & 3][2 * qp], e1 = qtab[t & 3][2 * qp + 1];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int4 e = h ? e1 : e0;
                const int yb = e.z >> 16, xb = e.z & 0xFFFF;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const bool ok = (e.w != 0) & (aoff[i] >= 0);
                    const unsigned off = (unsigned)(e.y + aoff[i]) * 4u;
                    ra[2 * i + h] = bload_f32(ars, ok ? off : 0xFFFFFFFFu);
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int iy = yb + (bpk[i] & 255) - 64, ix = xb + ((bpk[i] >> 8) & 255) - 64;
                    const bool ok = (e.w != 0) & ((bpk[i] >> 16) != 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    const unsigned off = (unsigned)(e.x + boff[i]) * 4u;
                    rb[2 * i + h] = bload_f32(brs, ok ? off : 0xFFFFFFFFu);
                }
            }
        };
        auto stash = [&](int buf, const float (&ra)[2 * RA], const float (&rb)[2 * RB]) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                unsigned hi, lo;
                const int off = (rg + NRG * i) * LDK + 2 * qp;
                split2(ra[2 * i], ra[2 * i + 1], hi, lo);
                *reinterpret_cast<unsigned*>(&Ahi[buf][off]) = hi;
                *reinterpret_cast<unsigned*>(&Alo[buf][off]) = lo;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                unsigned hi, lo;
                const int off = (rg + NRG * i) * LDK + 2 * qp;
                split2(rb[2 * i], rb[2 * i + 1], hi, lo);
                *reinterpret_cast<unsigned*>(&Bhi[buf][off]) = hi;
                *reinterpret_cast<unsigned*>(&Blo[buf][off]) = lo;
            }
        };
        float ra0[2 * RA], ra1[2 * RA], rb0[2 * RB], rb1[2 * RB];
        // stash tile t, refill its registers with tile t+2 (pipeline of conv_igemm_bf16x3_ws_kernel)
        build_tab(0);
        build_tab(1);
        build_tab(2);
        build_tab(3);
        __syncthreads();
        fetch(0, ra0, rb0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, ra1, rb1);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < nt; t += 2) {
            stash(0, ra0, rb0);
            build_tab(t + 4);
            fetch(t + 2, ra0, rb0);
            __syncthreads();
            stash(1, ra1, rb1);      // (no mid-pair exit: see conv_igemm_bf16x3_ws_kernel)
            build_tab(t + 5);
            fetch(t + 3, ra1, rb1);
            __syncthreads();
        }
        return;
    }

    const int cw = wave - NPW;
    const int wm0 = (cw / CWN) * (TM * 32), wn0 = (cw % CWN) * (TN * 32);
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 bhi[TN], blo[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = (wn0 + j * 32 + fr) * LDK + ks * 16 + 8 * fh;
                bhi[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                blo[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int o = (wm0 + i * 32 + fr) * LDK + ks * 16 + 8 * fh;
                const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
                const bf16x8 alo = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    __syncthreads();
    // the consumer skew of igemm_bf16x3_body: (row block 1, k-step 1) of a K-tile runs behind the next barrier
    constexpr bool SKEW = (BK / 16 == 2) && TM == 2 && (AG2V_CONSUMER_SKEW != 0);
    if constexpr (SKEW) {
        auto ldB = [&](int buf, int ks, bf16x8 (&bh)[TN], bf16x8 (&bl)[TN]) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = (wn0 + j * 32 + fr) * LDK + ks * 16 + 8 * fh;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
        };
        auto ldA = [&](int buf, int ks, int i, bf16x8& ah, bf16x8& al) {
            const int o = (wm0 + i * 32 + fr) * LDK + ks * 16 + 8 * fh;
            ah = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
            al = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
        };
        auto mm = [&](int i, const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[TN], const bf16x8 (&bl)[TN]) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
        };
        bf16x8 hbh[TN], hbl[TN], hah, hal;
        bool have = false;
        for (int t = 0; t < ((nt + 1) & ~1); ++t) {
            __syncthreads();   // tile t is in LDS buffer t & 1
            if (t < nt) {
                const int buf = t & 1;
                bf16x8 bh0[TN], bl0[TN], a0h, a0l, a1h, a1l;
                ldB(buf, 0, bh0, bl0);
                ldA(buf, 0, 0, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                if (have) mm(1, hah, hal, hbh, hbl);
                __builtin_amdgcn_sched_barrier(0);
                ldA(buf, 0, 1, a1h, a1l);
                mm(0, a0h, a0l, bh0, bl0);
                ldB(buf, 1, hbh, hbl);
                ldA(buf, 1, 0, a0h, a0l);
                mm(1, a1h, a1l, bh0, bl0);
                ldA(buf, 1, 1, hah, hal);
                mm(0, a0h, a0l, hbh, hbl);
                have = true;
            }
        }
        if (have) mm(1, hah, hal, hbh, hbl);
    } else
    for (int t = 0; t < ((nt + 1) & ~1); ++t) {   // the producers run an even number of iterations
        __syncthreads();   // tile t is in LDS buffer t & 1
        if (t < nt) compute(t & 1);
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int colj = n0 + wn0 + j * 32 + (lane & 31);
        if (colj >= p.NG) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float* dst = outg + (size_t)m * p.NG + colj;
                    if (atomic_out)
                        atomicAdd(dst, acc[i][j][r]);
                    else
                        *dst = acc[i][j][r];
                }
            }
        }
    }
}

// Weight gradient for stride-1 convolutions with OW % 8 == 0: the K dimension of this GEMM is the pixel index q, which is
// contiguous in memory for both operands (rows of dY; rows of x shifted by the tap), so a producer thread owns 8 consecutive
// pixels of one row: two 16-byte buffer loads (4-byte aligned for the shifted x rows), 4 pair splits, and one 16-byte LDS store
// for the hi and the lo image each -- against 8 dword loads and 8 ds_write_b32 in conv_wgrad_bf16x3_ws_kernel.  Only the first /
// last element of a run can fall outside the image (pad <= 8 - ... checked by the launcher: pad <= S - 1 <= 7); such runs are
// flagged per K-tile by the table builder and patched after the load.
#ifndef AG2V_WGRAD_LDK
#define AG2V_WGRAD_LDK 40      // LDS row stride in bf16 elements of the 16-byte-load weight-gradient kernels (experiments override it)
#endif
// F32: the same pipeline on EXACT fp32 operands (v_mfma_f32_32x32x2_f32; ConvArgs::f32): the producers store the words as loaded,
// [row][32 pixels] with 36-word rows; consumer lane (row, half) multiplies pixels 16 half + j (the order of the terms of a K-tile is
// free as long as both operands use the same): four 16-byte LDS reads per 32-row fragment, as in conv_igemm_f32_ws_kernel.
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16, bool PSA, bool PSX, bool F32 = false>
__device__ __forceinline__ void wgrad_bf16x3_v_body(const ConvArgs& p) {
    static_assert(!F32 || (!MF16 && !PSA && !PSX), "fp32 operands: 32x32 tiles on fp32 tensors");
    constexpr int BK = 32, LDK = AG2V_WGRAD_LDK, NP = NPW * 64;
    constexpr int LDKF = BK + 4;
    constexpr int TM = BM / CWM / 32, TN = BN / CWN / 32;
    constexpr int RPT = NP / 4;                       // rows covered by one pass of the producer threads (4 runs of 8 q per row)
    constexpr int NA = BM / RPT, NB = BN / RPT;       // A / B rows per producer thread
    static_assert(BM % RPT == 0 && BN % RPT == 0, "producer mapping");
    __shared__ __attribute__((aligned(16))) unsigned short Ahi[2][F32 ? 8 : BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Alo[2][F32 ? 8 : BM * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Bhi[2][F32 ? 8 : BN * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Blo[2][F32 ? 8 : BN * LDK];
    __shared__ __attribute__((aligned(16))) float Af[2][F32 ? BM * LDKF : 4];
    __shared__ __attribute__((aligned(16))) float Bf[2][F32 ? BN * LDKF : 4];
    __shared__ int4 qtab[4][4];   // per run of 8 pixels of a K-tile: {x offset of tap (0,0), dY offset, (yb<<16)|xb, flags}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < NPW;
    // 1-D grid over (pixel-range split, tile), consecutive ids on one XCD: the tiles of one split read the same dY rows (and
    // neighbouring x channels), so they share that XCD's L2 instead of fetching dY once per tile (PMC: 4.8 GB -> see profiles)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int gl = xcd_remap(blockIdx.x, ntiles * p.splitk);
    const int split = gl / ntiles;
    const int lid = gl - split * ntiles;
    const int tile_m = lid % p.tiles_m, tile_n = lid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = p.R * p.S;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    // pixel range of this workgroup: split sl of sample group grp (groups == 1: the whole batch is one group)
    const int grp = split / p.spg, sl = split - grp * p.spg;
    const int qs = grp * p.gq + sl * p.kchunk;
    const int qe = min(qs + p.kchunk, (grp + 1) * p.gq);
    if (qs >= qe) return;
    const int nt = (qe - qs + BK - 1) / BK;
    float* const outg = p.det ? p.out + (size_t)split * ((size_t)p.M * p.NG) : p.out + (size_t)grp * p.out_gstride;
    const bool atomic_out = !p.det && (p.spg > 1 || p.accumulate);

    if (producer) {
        const __amdgpu_buffer_rsrc_t ars = make_rsrc(p.a_src, p.a_bytes), brs = make_rsrc(p.b_src, p.b_bytes);
        const int run = tid & 3, row = tid >> 2;      // run of 8 pixels inside the K-tile; rows / columns row + RPT i
        // Validity without per-load selects where the GEMM allows it (producer VALU beside the prioritised MFMA waves costs ~1 %
        // of the kernel per instruction; a build without any validity logic ran 8-14 % faster): rows m >= M and columns j >= NG
        // only feed accumulators that are never stored, so they read row 0 / column 0; a run past the end of the split's pixel
        // range has byte offset 2^31 in the table for dY (the launcher takes this kernel for a_bytes <= 2^31 only), i.e. zeros,
        // which annihilates whatever x is read beside it.  What remains per x row is the image-row test of its tap.
        unsigned aoff4[NA];
        int boff[NB], bdy[NB], bdx[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + row + RPT * i;
            aoff4[i] = (unsigned)((m < p.M) ? m : 0) * (unsigned)OHW * 4u;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = n0 + row + RPT * i;
            int c, r, s;
            decode_crs((j < p.NG) ? j : 0, RS, p.S, c, r, s);
            bdy[i] = r - p.pad;
            bdx[i] = s - p.pad;
            boff[i] = c * HW + (r - p.pad) * p.W + (s - p.pad);
        }
        auto build_tab = [&](int t) {   // t = tile index relative to qs
            if (tid < 4) {
                const int q = qs + t * BK + tid * 8;
                const bool v = q < qe;   // Q % 8 == 0 and qs % 8 == 0: a run is entirely inside or outside the range
                const int qc = v ? q : qs;
                const int n = qc / OHW;
                const int pixl = qc - n * OHW;
                const int oy = pixl / p.OW, ox = pixl - oy * p.OW;
                // border run: some tap column reaches outside [0, W) for its first or last pixels
                const bool border = (ox - p.pad < 0) | (ox + 7 + (p.S - 1 - p.pad) >= p.W);
                // .x: element offset of the run in x (0 for a run past the range), .y: BYTE offset of the run in dY (2^31 past the
                // range), .w: bit 1 border run, bit 2 the run's first taps may lie before the tensor's first element
                const int xo = n * p.C * HW + oy * p.W + ox;
                qtab[t & 3][tid] = make_int4(v ? xo : 0, v ? (n * p.K * OHW + pixl) * 4 : (int)0x80000000u, (oy << 16) | ox,
                                             (v ? 1 : 0) | (border ? 2 : 0) | ((v && xo < p.pad * (p.W + 1)) ? 4 : 0));
            }
        };
        // fix word: bit (2 i) / (2 i + 1) set when B row i needs its leading / trailing elements patched (see stash)
        auto fetch = [&](int t, u32x4 (&ra)[NA][2], u32x4 (&rb)[NB][2], int& fixw, int& fixx, int& shw) {
            const int4 e = qtab[t & 3][run];
            asm volatile("" ::: "memory");
            const int yb = e.z >> 16, xb = e.z & 0xFFFF;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned off = (unsigned)e.y + aoff4[i];
                ra[i][0] = bload_u32x4(ars, off);
                ra[i][1] = bload_u32x4(ars, off + 16u);
            }
            fixw = (e.w & 2) ? 1 : 0;
            fixx = xb;
            // Every row takes the same two 16-byte loads: a branch around loads inside this pipelined loop makes the compiler's
            // vmcnt bookkeeping pessimistic (it waited down to vmcnt(3), i.e. for the loads issued half an iteration ago; a build
            // without the branch ran 8-14 % faster).  The one case that needs other addresses -- first rows of image 0, where a
            // run starts up to `pad` elements before the tensor's first element and a wrapped offset would read 0 in all four
            // dwords -- loads from element 0 instead and shifts the registers in stash (shw: 2 bits per row).
            int o[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) o[i] = e.x + boff[i];
            shw = 0;
            if (__builtin_expect((e.w & 4) != 0, 0)) {
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (o[i] < 0) {      // (a row whose tap row is outside the image is not loaded at all: no shift recorded)
                        if ((unsigned)(yb + bdy[i]) < (unsigned)p.H) shw |= ((-o[i]) & 3) << (2 * i);
                        o[i] = 0;
                    }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int iy = yb + bdy[i];
                const unsigned off = ((unsigned)iy < (unsigned)p.H) ? (unsigned)o[i] * 4u : 0xFFFFFFE0u;
                rb[i][0] = bload_u32x4(brs, off);
                rb[i][1] = bload_u32x4(brs, off + 16u);
            }
        };
        auto store_row = [&](auto ps, unsigned short* hi_img, unsigned short* lo_img, int r, const u32x4 (&v)[2]) {
            constexpr bool PS = decltype(ps)::value;
            unsigned h[4], l[4];
            split_or_regroup<PS>(__uint_as_float(v[0][0]), __uint_as_float(v[0][1]), h[0], l[0]);
            split_or_regroup<PS>(__uint_as_float(v[0][2]), __uint_as_float(v[0][3]), h[1], l[1]);
            split_or_regroup<PS>(__uint_as_float(v[1][0]), __uint_as_float(v[1][1]), h[2], l[2]);
            split_or_regroup<PS>(__uint_as_float(v[1][2]), __uint_as_float(v[1][3]), h[3], l[3]);
            const int off = r * LDK + 8 * run;
            *reinterpret_cast<uint4*>(&hi_img[off]) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(&lo_img[off]) = make_uint4(l[0], l[1], l[2], l[3]);
        };
        auto store_row_f = [&](float* img, int r, const u32x4 (&v)[2]) {      // fp32 words as loaded
            *reinterpret_cast<u32x4*>(&img[r * LDKF + 8 * run]) = v[0];
            *reinterpret_cast<u32x4*>(&img[r * LDKF + 8 * run + 4]) = v[1];
        };
        auto stash = [&](int buf, const u32x4 (&ra)[NA][2], u32x4 (&rb)[NB][2], int fixw, int fixx, int shw) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if constexpr (F32) store_row_f(Af[buf], row + RPT * i, ra[i]);
                else store_row(std::integral_constant<bool, PSA>{}, Ahi[buf], Alo[buf], row + RPT * i, ra[i]);
            }
            if (__builtin_expect(shw != 0, 0)) {   // rows loaded from element 0 instead of element -sh: move them up by sh places
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int sh = (shw >> (2 * i)) & 3;
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (sh > q) {
                            rb[i][1] = u32x4{rb[i][0][3], rb[i][1][0], rb[i][1][1], rb[i][1][2]};
                            rb[i][0] = u32x4{0u, rb[i][0][0], rb[i][0][1], rb[i][0][2]};
                        }
                }
            }
            if (fixw) {   // (almost) wave-uniform: the 4 runs of a K-tile are 32 consecutive pixels
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int ix0 = fixx + bdx[i];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if ((unsigned)(ix0 + k) >= (unsigned)p.W) rb[i][k >> 2][k & 3] = 0u;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if constexpr (F32) store_row_f(Bf[buf], row + RPT * i, rb[i]);
                else store_row(std::integral_constant<bool, PSX>{}, Bhi[buf], Blo[buf], row + RPT * i, rb[i]);
            }
        };
        u32x4 ra0[NA][2], ra1[NA][2], rb0[NB][2], rb1[NB][2];
        int fw0, fx0, fw1, fx1, sh0, sh1;
        build_tab(0);
        build_tab(1);
        build_tab(2);
        build_tab(3);
        __syncthreads();
        fetch(0, ra0, rb0, fw0, fx0, sh0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, ra1, rb1, fw1, fx1, sh1);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < nt; t += 2) {       // pipeline of conv_igemm_bf16x3_ws_kernel
            stash(0, ra0, rb0, fw0, fx0, sh0);
            build_tab(t + 4);
            fetch(t + 2, ra0, rb0, fw0, fx0, sh0);
            __syncthreads();
            stash(1, ra1, rb1, fw1, fx1, sh1);
            build_tab(t + 5);
            fetch(t + 3, ra1, rb1, fw1, fx1, sh1);
            __syncthreads();
        }
        return;
    }

    const int cw = wave - NPW;
    if (!p.diag_noprio) __builtin_amdgcn_s_setprio(2);  // MFMA waves first on the SIMD's vector-issue port (see the igemm kernel): -0.45 % step
    const int wm0 = (cw / CWN) * (TM * 32), wn0 = (cw % CWN) * (TN * 32);
    // MFMA shape: see conv_igemm_bf16x3_ws_kernel
    constexpr int EM = MF16 ? TM * 2 : TM, EN = MF16 ? TN * 2 : TN;
    constexpr int ER = MF16 ? 16 : 32, EREGS = MF16 ? 4 : 16;
    typedef float accv __attribute__((ext_vector_type(EREGS)));
    accv acc[EM][EN];
#pragma unroll
    for (int i = 0; i < EM; ++i)
#pragma unroll
        for (int j = 0; j < EN; ++j)
#pragma unroll
            for (int r = 0; r < EREGS; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & (ER - 1), fq = lane / ER;
    auto compute = [&](int buf) {
        if constexpr (F32) {
            const int half = lane >> 5, r32 = lane & 31;
            float4 bfr[EN][4];
#pragma unroll
            for (int j = 0; j < EN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bfr[j][q] = *reinterpret_cast<const float4*>(&Bf[buf][(wn0 + j * 32 + r32) * LDKF + 16 * half + 4 * q]);
#pragma unroll
            for (int i = 0; i < EM; ++i) {
                float4 afr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    afr[q] = *reinterpret_cast<const float4*>(&Af[buf][(wm0 + i * 32 + r32) * LDKF + 16 * half + 4 * q]);
#pragma unroll
                for (int j = 0; j < EN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].x, bfr[j][q].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].y, bfr[j][q].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].z, bfr[j][q].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q].w, bfr[j][q].w, acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < (MF16 ? 1 : BK / 16); ++ks) {
            bf16x8 bhi[EN], blo[EN];
#pragma unroll
            for (int j = 0; j < EN; ++j) {
                const int o = (wn0 + j * ER + fr) * LDK + ks * 16 + 8 * fq;
                bhi[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                blo[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
#pragma unroll
            for (int i = 0; i < EM; ++i) {
                const int o = (wm0 + i * ER + fr) * LDK + ks * 16 + 8 * fq;
                const bf16x8 ahi = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
                const bf16x8 alo = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
#pragma unroll
                for (int j = 0; j < EN; ++j) {
                    if constexpr (MF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bhi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, blo[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bhi[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };
    __syncthreads();
    // the consumer skew of igemm_bf16x3_body: (row block 1, k-step 1) of a K-tile runs behind the next barrier
    constexpr bool SKEW = !MF16 && !F32 && (BK / 16 == 2) && EM == 2 && (AG2V_CONSUMER_SKEW != 0);
    if constexpr (SKEW) {
        auto ldB = [&](int buf, int ks, bf16x8 (&bh)[EN], bf16x8 (&bl)[EN]) {
#pragma unroll
            for (int j = 0; j < EN; ++j) {
                const int o = (wn0 + j * ER + fr) * LDK + ks * 16 + 8 * fq;
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bhi[buf][o]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Blo[buf][o]);
            }
        };
        auto ldA = [&](int buf, int ks, int i, bf16x8& ah, bf16x8& al) {
            const int o = (wm0 + i * ER + fr) * LDK + ks * 16 + 8 * fq;
            ah = *reinterpret_cast<const bf16x8*>(&Ahi[buf][o]);
            al = *reinterpret_cast<const bf16x8*>(&Alo[buf][o]);
        };
        auto mm = [&](int i, const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[EN], const bf16x8 (&bl)[EN]) {
#pragma unroll
            for (int j = 0; j < EN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
        };
        bf16x8 hbh[EN], hbl[EN], hah, hal;
        bool have = false;
        for (int t = 0; t < ((nt + 1) & ~1); ++t) {
            __syncthreads();   // tile t is in LDS buffer t & 1
            if (t < nt) {
                const int buf = t & 1;
                bf16x8 bh0[EN], bl0[EN], a0h, a0l, a1h, a1l;
                ldB(buf, 0, bh0, bl0);
                ldA(buf, 0, 0, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                if (have) mm(1, hah, hal, hbh, hbl);
                __builtin_amdgcn_sched_barrier(0);
                ldA(buf, 0, 1, a1h, a1l);
                mm(0, a0h, a0l, bh0, bl0);
                ldB(buf, 1, hbh, hbl);
                ldA(buf, 1, 0, a0h, a0l);
                mm(1, a1h, a1l, bh0, bl0);
                ldA(buf, 1, 1, hah, hal);
                mm(0, a0h, a0l, hbh, hbl);
                have = true;
            }
        }
        if (have) mm(1, hah, hal, hbh, hbl);
    } else
    for (int t = 0; t < ((nt + 1) & ~1); ++t) {   // the producers run an even number of iterations
        __syncthreads();   // tile t is in LDS buffer t & 1
        if (t < nt) compute(t & 1);
    }

#pragma unroll
    for (int j = 0; j < EN; ++j) {
        const int colj = n0 + wn0 + j * ER + (lane & (ER - 1));
        if (colj >= p.NG) continue;
#pragma unroll
        for (int i = 0; i < EM; ++i) {
#pragma unroll
            for (int r = 0; r < EREGS; ++r) {
                const int m = MF16 ? m0 + wm0 + i * 16 + 4 * (lane >> 4) + r
                                   : m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float* dst = outg + (size_t)m * p.NG + colj;
                    if (atomic_out)
                        atomicAdd(dst, acc[i][j][r]);
                    else
                        *dst = acc[i][j][r];
                }
            }
        }
    }
}
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_wgrad_bf16x3_v_kernel(ConvArgs p) {
    wgrad_bf16x3_v_body<BM, BN, CWM, CWN, NPW, MF16, false, false>(p);
}
// dy (PSA: ConvArgs::a_split) and / or x (PSX: ConvArgs::b_split) hold split words
// exact fp32 on the same pipeline (ConvArgs::f32): conv_wgrad_kernel's arithmetic with 16-byte loads and 8 + 8 waves
template <int BM, int BN, int CWM, int CWN, int NPW>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_wgrad_f32_v_kernel(ConvArgs p) {
    wgrad_bf16x3_v_body<BM, BN, CWM, CWN, NPW, false, false, false, true>(p);
}
template <int BM, int BN, int CWM, int CWN, int NPW, bool MF16, bool PSA, bool PSX>
__global__ __launch_bounds__((NPW + CWM * CWN) * 64) void conv_wgrad_bf16x3_vps_kernel(ConvArgs p) {
    wgrad_bf16x3_v_body<BM, BN, CWM, CWN, NPW, MF16, PSA, PSX>(p);
}

// ================================================================ launchers
// One 16-wave workgroup occupies a CU (124 KB LDS), so a launch runs in rounds of 256 workgroups.  Pick the number of K-range
// splits that minimises  rounds * (K-tiles per split + overhead)  -- overhead ~ 8 K-tiles for the exposed first loads, the
// epilogue and (with splits) its atomics -- instead of a fixed "fill the chip" factor that can leave a nearly empty last round.
static int pick_splits(int tiles, int nk, int max_splits) {
    constexpr int forced = 0;
    if (forced > 0) return forced < max_splits ? forced : (max_splits > 0 ? max_splits : 1);
    int best = 1;
    long long best_cost = -1;
    for (int sk = 1; sk <= max_splits; ++sk) {
        const int chunk = ceil_div(nk, sk);
        if (sk > 1 && ceil_div(nk, chunk) != sk) continue;          // same chunking as a smaller split count
        const long long rounds = ceil_div(tiles * sk, 256);
        const long long cost = rounds * (chunk + (sk > 1 ? 12 : 8));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sk; }
    }
    return best;
}

int launch_igemm_bf16x3(ConvArgs& a, hipStream_t st, bool out_zeroed) {
    if (a.o_mul == 0) { a.o_mul = 1; a.o_add_y = 0; a.o_add_x = 0; a.o_H = a.OH; a.o_W = a.OW; }      // (direct callers: the dense output)
    if (a.out_nstride == 0) a.out_nstride = (long long)a.M * a.o_H * a.o_W;
    if (igemm_halo_ok(a)) return launch_igemm_halo(a, st);       // 3x3 stride-1 on split words: halo tiles by LDS-DMA (conv_halo.hip)
    // 256 x 128 tile when there are enough rows and it pads no more rows than the 128 x 256 tile does (384 rows: three 128-row
    // tiles, not two 256-row ones)
    const bool tall = a.M >= 192 && ceil_div(a.M, 256) * 256 <= ceil_div(a.M, 128) * 128;
    const int bm = tall ? 256 : 128, bn = tall ? 128 : 256;
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = ceil_div(a.NG, bn);
    const int tiles = a.tiles_m * a.tiles_n, nk = ceil_div(a.KG, 32);
    a.splitk = 1; a.kchunk = nk;
    const bool dense_out = a.out_nstride == (long long)a.M * a.o_H * a.o_W;      // (a strided output is written once, by plain stores)
    if (tiles < 1024 && nk >= 8 && !a.det && dense_out) {      // (deterministic: one workgroup owns an output tile's whole reduction)
        const int sk = pick_splits(tiles, nk, nk / 4 < 64 ? nk / 4 : 64);
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
#define AG2V_STR_(...) #__VA_ARGS__
#define AG2V_STR(...) AG2V_STR_(__VA_ARGS__)
#define AG2V_LAUNCH_IGEMM(...)                                                                          \
    do {                                                                                               \
        if (a.b_split) {                                                                               \
            set_last_conv_kernel("conv_igemm_bf16x3_ps_kernel<" AG2V_STR(__VA_ARGS__) ">");            \
            hipLaunchKernelGGL((conv_igemm_bf16x3_ps_kernel<__VA_ARGS__>), grid, dim3(1024), 0, st, a); \
        } else {                                                                                       \
            set_last_conv_kernel("conv_igemm_bf16x3_ws_kernel<" AG2V_STR(__VA_ARGS__) ">");            \
            hipLaunchKernelGGL((conv_igemm_bf16x3_ws_kernel<__VA_ARGS__>), grid, dim3(1024), 0, st, a); \
        }                                                                                              \
    } while (0)
    // quad loads of the gathered operand on the 128 x 256 tile: stride-1 launches whose pixel rows are whole quads
    // (AG2V_NO_B4: A/B aid).  The 256 x 128 tile (8 of its 12 loads gather) was 4 % slower with them.
    static const bool b4_off = false;
    static const bool b4_tall = false;
    const bool b4 = a.tap_major && !b4_off && (!tall || b4_tall) && a.stride == 1 && a.OW % 4 == 0 && a.W >= 4 &&
                    (a.OH * a.OW) % 4 == 0 && a.pad_w <= 3;
    if (a.h16) {        // fp16 split (weights packed by the *_h kernels)
        if (a.b_split || a.out_split || a.f32) return AG2V_EINVAL;
#define AG2V_LAUNCH_H16(...)                                                                           \
    do {                                                                                               \
        set_last_conv_kernel("conv_igemm_f16x3_ws_kernel<" AG2V_STR(__VA_ARGS__) ">");                 \
        hipLaunchKernelGGL((conv_igemm_f16x3_ws_kernel<__VA_ARGS__>), grid, dim3(1024), 0, st, a);      \
    } while (0)
        if (tall) {
            if (b4) AG2V_LAUNCH_H16(256, 128, 4, 2, 8, true, true);
            else if (a.tap_major) AG2V_LAUNCH_H16(256, 128, 4, 2, 8, true, false);
            else AG2V_LAUNCH_H16(256, 128, 4, 2, 8, false, false);
        } else {
            if (b4) AG2V_LAUNCH_H16(128, 256, 2, 4, 8, true, true);
            else if (a.tap_major) AG2V_LAUNCH_H16(128, 256, 2, 4, 8, true, false);
            else AG2V_LAUNCH_H16(128, 256, 2, 4, 8, false, false);
        }
#undef AG2V_LAUNCH_H16
    } else if (a.f32) {        // exact fp32 operands (k-major fp32 weights, table path)
        if (a.tap_major || a.b_split || a.out_split) return AG2V_EINVAL;
        if (tall) {
            set_last_conv_kernel("conv_igemm_f32_ws_kernel<256, 128, 4, 2, 8>");
            hipLaunchKernelGGL((conv_igemm_f32_ws_kernel<256, 128, 4, 2, 8>), grid, dim3(1024), 0, st, a);
        } else {
            set_last_conv_kernel("conv_igemm_f32_ws_kernel<128, 256, 2, 4, 8>");
            hipLaunchKernelGGL((conv_igemm_f32_ws_kernel<128, 256, 2, 4, 8>), grid, dim3(1024), 0, st, a);
        }
    } else if (tall) {
        // 16x16x32 MFMAs (+3..6 % with the table-path producers) read their LDS operands with 36 % bank-conflict cycles
        // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, tools/ubench/pmc_variants.sh; 32x32x16: 0 %) and leave the prioritised
        // MFMA waves' partners 8 instead of 24 free issue cycles per MFMA: with the TAPK producers the 32x32x16 shape is
        // 0.6 % of a c3 step faster (AG2V_TALL16 selects the other one for A/B runs).
        static const bool tall16 = false;
        if (b4) AG2V_LAUNCH_IGEMM(256, 128, 4, 2, 8, false, true, true);
        else if (a.tap_major && !tall16) AG2V_LAUNCH_IGEMM(256, 128, 4, 2, 8, false, true, false);
        else if (a.tap_major) AG2V_LAUNCH_IGEMM(256, 128, 4, 2, 8, AG2V_TALL_MF16, true, false);
        else AG2V_LAUNCH_IGEMM(256, 128, 4, 2, 8, AG2V_TALL_MF16, false, false);
    } else {
        if (b4) AG2V_LAUNCH_IGEMM(128, 256, 2, 4, 8, false, true, true);
        else if (a.tap_major) AG2V_LAUNCH_IGEMM(128, 256, 2, 4, 8, AG2V_WIDE_MF16, true, false);
        else AG2V_LAUNCH_IGEMM(128, 256, 2, 4, 8, false, false, false);
    }
#undef AG2V_LAUNCH_IGEMM
    AG2V_CHECK_LAUNCH();
    if (a.splitk > 1 && (a.relu || a.mask || a.out_split))
        return launch_act_mask(a.out, a.mask, (long long)a.N * a.M * a.o_H * a.o_W, a.relu, st, a.out_split);
    return AG2V_OK;
}

void plan_wgrad_bf16x3(ConvArgs& a) {
    if (wgrad_halo_ok(a)) { plan_wgrad_halo(a); return; }
    const bool tall = a.M >= 192 && ceil_div(a.M, 256) * 256 <= ceil_div(a.M, 128) * 128;     // see launch_igemm_bf16x3
    const int bm = tall ? 256 : 128, bn = tall ? 128 : 256;
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = ceil_div(a.NG, bn);
    const int tiles = a.tiles_m * a.tiles_n;
    // per sample group (a.groups == 1: the whole batch): K-tiles of 32 pixels, spg pixel-range splits
    if (a.groups < 1) { a.groups = 1; a.gq = a.KG; a.out_gstride = (long long)a.M * a.NG; }
    const long long Qg = a.gq;
    const int nq = (int)((Qg + 31) / 32);
    int spg = pick_splits(tiles * a.groups, nq, nq / 8 < 256 ? (nq / 8 > 0 ? nq / 8 : 1) : 256);
    int kchunk = ceil_div(nq, spg) * 32;
    spg = (int)((Qg + kchunk - 1) / kchunk);
    a.spg = spg; a.splitk = spg * a.groups; a.kchunk = kchunk;
}

// exact-fp32 weight gradients that the 16-byte-load pipeline takes (the others stay on conv_wgrad_kernel): stride 1, pixel rows of whole
// 8-pixel runs, at least 128 rows (fewer rows: the 4-wave kernels' 32- / 64-row tiles pad less), dY within the 2^31-byte offset encoding
bool wgrad_f32_ws_ok(const ConvArgs& a) {
    static const bool off = false;
    return !off && a.M >= 128 && a.stride == 1 && a.OW % 8 == 0 && a.W >= 8 && a.pad <= 3 && a.a_bytes != 0 && a.a_bytes <= 0x80000000u &&
           ((long long)a.N * a.OH * a.OW) % 8 == 0;
}

int launch_wgrad_bf16x3(ConvArgs& a, hipStream_t st) {
    if (wgrad_halo_ok(a)) return launch_wgrad_halo(a, st);
    plan_wgrad_bf16x3(a);
    const bool tall = a.M >= 192 && ceil_div(a.M, 256) * 256 <= ceil_div(a.M, 128) * 128;
    const int tiles = a.tiles_m * a.tiles_n;
    const int spg = a.spg, splitk = a.splitk;
    if (spg > 1 && !a.accumulate && !a.det) {
        if (hipMemsetAsync(a.out, 0, (size_t)a.M * a.NG * a.groups * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    const dim3 grid(tiles * splitk);     // 1-D: the kernels decode (split, tile) themselves (XCD-aware)
    // 32x32x16 MFMAs: the 16x16x32 operand reads cost 49 % LDS bank-conflict cycles in these kernels (pmc_variants.sh) and the
    // step is 0.35 % faster without them (AG2V_WGRAD16 selects the 16x16x32 form for A/B runs)
    static const bool wgrad32 = true;
    static const int wnoprio = 0;
    a.diag_noprio = wnoprio;
    if (a.f32) {        // exact fp32 operands: the 16-byte-load kernels only (wgrad_f32_ws_ok)
        if (!wgrad_f32_ws_ok(a) || a.a_split || a.b_split) return AG2V_EINVAL;
        if (tall) {
            set_last_conv_kernel("conv_wgrad_f32_v_kernel<256, 128, 4, 2, 8>");
            hipLaunchKernelGGL((conv_wgrad_f32_v_kernel<256, 128, 4, 2, 8>), grid, dim3(1024), 0, st, a);
        } else {
            set_last_conv_kernel("conv_wgrad_f32_v_kernel<128, 256, 2, 4, 8>");
            hipLaunchKernelGGL((conv_wgrad_f32_v_kernel<128, 256, 2, 4, 8>), grid, dim3(1024), 0, st, a);
        }
        AG2V_CHECK_LAUNCH();
        return AG2V_OK;
    }
    if (a.stride == 1 && a.OW % 8 == 0 && a.W >= 8 && a.a_bytes <= 0x80000000u && a.pad <= 3 &&
        true) {
        // 16-byte loads along the pixel index (runs of 8 pixels never straddle an image row)
        if (a.b_split || a.a_split) {       // dy and / or x as split words: the 32x32x16 forms
#define AG2V_VPS(BM_, BN_, WM_, WN_, PA_, PX_)                                                                                   \
            do {                                                                                                              \
                set_last_conv_kernel("conv_wgrad_bf16x3_vps_kernel<" #BM_ ", " #BN_ ", " #WM_ ", " #WN_ ", 8, false, " #PA_ ", " #PX_ ">"); \
                hipLaunchKernelGGL((conv_wgrad_bf16x3_vps_kernel<BM_, BN_, WM_, WN_, 8, false, PA_, PX_>), grid, dim3(1024), 0, st, a); \
            } while (0)
            if (tall) {
                if (a.a_split && a.b_split) AG2V_VPS(256, 128, 4, 2, true, true);
                else if (a.a_split) AG2V_VPS(256, 128, 4, 2, true, false);
                else AG2V_VPS(256, 128, 4, 2, false, true);
            } else {
                if (a.a_split && a.b_split) AG2V_VPS(128, 256, 2, 4, true, true);
                else if (a.a_split) AG2V_VPS(128, 256, 2, 4, true, false);
                else AG2V_VPS(128, 256, 2, 4, false, true);
            }
#undef AG2V_VPS
        } else if (tall) {
            if (wgrad32) {
                set_last_conv_kernel("conv_wgrad_bf16x3_v_kernel<256, 128, 4, 2, 8, false>");
                hipLaunchKernelGGL((conv_wgrad_bf16x3_v_kernel<256, 128, 4, 2, 8, false>), grid, dim3(1024), 0, st, a);
            } else {
            set_last_conv_kernel("conv_wgrad_bf16x3_v_kernel<256, 128, 4, 2, 8, true>");
            hipLaunchKernelGGL((conv_wgrad_bf16x3_v_kernel<256, 128, 4, 2, 8, true>), grid, dim3(1024), 0, st, a);
            }
        } else {
            if (wgrad32) {
                set_last_conv_kernel("conv_wgrad_bf16x3_v_kernel<128, 256, 2, 4, 8, false>");
                hipLaunchKernelGGL((conv_wgrad_bf16x3_v_kernel<128, 256, 2, 4, 8, false>), grid, dim3(1024), 0, st, a);
            } else {
            set_last_conv_kernel("conv_wgrad_bf16x3_v_kernel<128, 256, 2, 4, 8, true>");
            hipLaunchKernelGGL((conv_wgrad_bf16x3_v_kernel<128, 256, 2, 4, 8, true>), grid, dim3(1024), 0, st, a);
            }
        }
    } else {
        if (a.b_split || a.a_split) return AG2V_EINVAL;      // only the 16-byte-load kernels read split words
        if (tall) {
            set_last_conv_kernel("conv_wgrad_bf16x3_ws_kernel<256, 128, 4, 2, 8>");
            hipLaunchKernelGGL((conv_wgrad_bf16x3_ws_kernel<256, 128, 4, 2, 8>), grid, dim3(1024), 0, st, a);
        } else {
            set_last_conv_kernel("conv_wgrad_bf16x3_ws_kernel<128, 256, 2, 4, 8>");
            hipLaunchKernelGGL((conv_wgrad_bf16x3_ws_kernel<128, 256, 2, 4, 8>), grid, dim3(1024), 0, st, a);
        }
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
