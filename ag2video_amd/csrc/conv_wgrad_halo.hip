// Weight gradient of a 3x3 stride-1 convolution on the bf16 matrix pipe ("bf16x3", conv_bf16x3.hip) with BOTH operands brought from memory
// to LDS by LDS-DMA (buffer_load ... lds) -- no producer waves.  gfx950 only.  Replaces, for the shapes it covers, the wave-specialised
// conv_wgrad_bf16x3_vps_kernel: the weight gradients of the SPADE decoder's mlp_shared / gamma | beta / conv_0 / conv_1 layers (reference
// models/spade_models/networks/normalization.py:90-94, architecture.py:29-32), whose operands the step hands over as SPLIT WORDS
// ((bf16 hi << 16) | bf16 lo per value).
//
//   dW[m][c][r][s] = sum over (n, y, x) of dY[n][m][y][x] * X[n][c][y + r - 1][x + s - 1]
//
// is a GEMM whose reduction runs over PIXELS: an MFMA operand fragment is 8 consecutive pixels of one dY channel (rows) / of one x channel
// at one filter tap (columns).  The 16-byte-load kernels build every column's fragments by themselves: 9 taps x C channels, each tap
// re-fetching the same pixels shifted by one, regrouped (split word -> hi / lo bf16 planes) and stored to LDS by 8 producer waves beside 8
// multiplying waves -- both sides saturated.  Here:
//   * a workgroup (8 waves, all of them multiply) owns 128 dY channels x (64 x channels x 9 taps) and walks a COLUMN STRIP of 32 pixels
//     down the image, one image row per step (a 32-deep k-step: 108 MFMAs 16x16x32 per wave);
//   * dY row segments [128][32 words] and x row segments [64][32 + 8 words] arrive by LDS-DMA (1 KiB pieces, per-lane source offsets computed
//     once, a step only moves a wave-uniform SGPR offset); image borders, channels beyond M and the zero rows above / below an image cost
//     nothing: their lanes carry an out-of-range offset, for which the DMA writes zeros;
//   * every x word is regrouped ONCE per step -- not once per tap and per multiplying wave: each thread turns 10 words of one (channel, octet)
//     into the three column-shifted hi (or lo) bf16 octets (12 v_perm, 3 ds_write_b128) of a 3-row ring of PLANES, from which all nine taps
//     of three consecutive steps read their B fragments as plain 16-byte reads;
//   * the dY fragments (4 per wave and step, reused by all 9 taps) are regrouped by the wave that multiplies them (32 v_perm per step),
//     one step ahead, into the other of two register sets.
// One s_barrier and one counted s_waitcnt vmcnt per step; DMA runs two steps ahead (3 raw buffers per operand).
// A sample group's strips form ONE stream of ring rows (every strip's rows behind a zero row); a workgroup owns a run [g0, g1) of it,
// which may begin and end inside a strip: the rows around the run are loaded as tap rows (x) but contribute no dY.  The runs of a launch
// are added through slabs + an ordered reduce pass when the caller hands over the workspace (conv_igemm.hip wgrad_launch), else by
// fp32 atomics.  AS / XS = false: that operand is fp32 and the regrouping step performs the hi / lo decomposition (split2) itself.
//
// Step jg (ring row jg: a strip's rows preceded by ONE zero row, which also is the previous strip's bottom border):
//     taps r = 0 (plane slot of row jg - 1)                                     36 MFMAs
//     wait: this wave's DMA pieces of step jg - 2 have landed; barrier          -> x raw (jg + 2), dY raw (jg + 1) certified for everyone;
//                                                                                  every wave is done with plane slot (jg - 1) % 3
//     issue DMA: x raw row jg + 4, dY raw row jg + 3
//     regroup x raw row jg + 2 -> plane slot (jg + 2) % 3 = (jg - 1) % 3
//     taps r = 1 (row jg), dY fragments of row jg + 1 -> the other register set, taps r = 2 (row jg + 1)          72 MFMAs
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_vptr;

namespace wgh {
constexpr int RM = 4;                               // 16-row blocks per wave
constexpr int BM = 2 * RM * 16, CB = 64, TW = 32;   // dY channels / x channels per workgroup, pixels per step
constexpr int A_BUF = BM * TW * 4;                  // dY raw buffer: [row][8 quads], quad' = quad ^ wgh_aswz(row)
constexpr int XR_ROW = (TW + 8) * 4;                // x raw: [channel][10 quads] = pixels x0 - 4 .. x0 + 35
constexpr int XR_BUF = CB * XR_ROW;
constexpr int PL_BLK = CB * TW * 2 + 128;           // one (shift, plane) block [channel][4 octets of 8 bf16]; + 128 B: consecutive blocks start
                                                    // 8 bank-quads apart, so the hi and lo writers of a channel's octets do not collide
constexpr int PL_SLOT = 6 * PL_BLK;
constexpr int A_OFF = 0, XR_OFF = 3 * A_BUF, PL_OFF = XR_OFF + 3 * XR_BUF, DUMMY_OFF = PL_OFF + 3 * PL_SLOT, LDS_BYTES = DUMMY_OFF + 1024;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr unsigned kOut = 0xFFFFFFF0u;              // an out-of-range buffer offset: the DMA writes zeros for that lane
}

#if defined(__HIP_DEVICE_COMPILE__)
// Quad swizzle of the dY raw rows (8 quads of 16 bytes per row, two rows per 256-byte bank row): a 16-lane group of a ds_read_b128 is
// {rows 0-3, 12-15 at k group g} + {rows 4-11 at k group g + 1}, whose quads differ by 2.  With f = 2 idx + [row in 4..11] (idx: the
// position of row / 2 among the four row pairs of its set) the first set lands on even quads, the second on odd ones: 16 distinct
// bank quads.  (quad ^ ((row >> 1) & 7), the first form, was 2-way on every read: SQ_LDS_BANK_CONFLICT 37 % of the LDS cycles.)
__device__ __forceinline__ int wgh_aswz(int row) {
    const int t = (row >> 1) & 7;
    const int in1 = ((t + 2) >> 2) & 1;
    const int idx = in1 ? (t - 2) : ((t & 1) | ((t >> 2) << 1));
    return 2 * idx + in1;
}
template <int N>
__device__ __forceinline__ void wgh_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// AS / XS: dY / x hold split words (false: fp32 -- the regrouping step then also performs the 3-term split's hi / lo decomposition)
template <bool AS, bool XS>
struct WgradHaloBody {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const ConvArgs& p;
    char* smem;
    __amdgpu_buffer_rsrc_t ars, brs;
    int wave, lane;
    // ---- DMA: per-lane source offsets (without the wave-uniform row offset)
    unsigned vA[2];                  // dY pieces wave, wave + 8: (m0 + row) * HW * 4 + quad * 16   (kOut: row beyond M)
    unsigned vX[2];                  // x pieces wave, wave + 8: c * HW * 4 + qq * 16               (piece >= 10: dummy)
    unsigned xedge[2];               // bit 0: this lane's quad is the left halo quad (qq == 0), bit 1: the right one (qq == 9)
    bool dummyX1;                    // the wave's second x piece does not exist (pieces 10..15)
    // ---- LDS addresses
    unsigned a_q0, a_q1;             // dY raw: row (lane & 15) of the wave's 64 rows, quads 2 g / 2 g + 1 (swizzled)
    unsigned pl_rd[3];               // planes, per ring slot: channel of this lane, octet g (swizzled)
    unsigned xr_rd, pl_wr, psel;     // regroup unit of this thread: raw row / plane destination / v_perm selector (hi or lo plane)
    // ---- row stream (wave-uniform)
    int H, W, HW, M, C, c0, m0, spr; // spr: strips per image row
    // A workgroup owns ring rows [g0, g1) of its sample group's row stream (strips x (H + 1) rows: every strip's rows behind one zero
    // row).  Its cursor starts one row earlier and runs past g1: the neighbouring rows' x is needed (as tap rows), their dY is not.
    struct Row { int data, own; unsigned offA, offX; int left, right; };      // data: a real image row; own: its dY belongs to this workgroup
    Row rows[5];                     // ring rows jg .. jg + 4
    int cst, cy, cn, cxs, cr;        // cursor of the row entering at jg + 4: strip number, y (-1: the zero row), image, strip column, ring row
    int g0, g1, nst_g;               // owned ring rows, strips of the group
    // ---- fragments
    bf16x8 ah[2][wgh::RM], al[2][wgh::RM];
    f32x4v acc[wgh::RM][9];

    __device__ __forceinline__ WgradHaloBody(const ConvArgs& p_, char* s) : p(p_), smem(s) {}

    // the row the cursor points at, then advance it
    __device__ __forceinline__ Row next_row() {
        Row r;
        if (cr < 0) {                 // (the row in front of the group's first strip)
            ++cr;
            r.data = 0; r.own = 0; r.offA = 0u; r.offX = 0u; r.left = 0; r.right = 0;
            return r;
        }
        r.data = (cst < nst_g && cy >= 0) ? 1 : 0;
        r.own = (r.data && cr >= g0 && cr < g1) ? 1 : 0;
        ++cr;
        const int x0 = cxs * wgh::TW;
        r.left = (x0 == 0) ? 1 : 0;
        r.right = (x0 + wgh::TW == W) ? 1 : 0;
        const unsigned pix = (unsigned)(cy * W + x0);
        r.offA = ((unsigned)(cn * M) * (unsigned)HW + pix) * 4u;
        r.offX = ((unsigned)(cn * C + c0) * (unsigned)HW + pix) * 4u - (r.left ? 0u : 16u);
        if (++cy == H) {
            cy = -1;
            ++cst;
            if (++cxs == spr) { cxs = 0; ++cn; }
        }
        return r;
    }
    // ---- the pieces of a step, issued in the shadow of its MFMAs (filler<T, I>: behind the three MFMAs of row block I of tap T)
    template <int BUF, int K>
    __device__ __forceinline__ void dma_a(const Row& r) {
        const unsigned v = r.own ? vA[K] : wgh::kOut;           // (rows of other workgroups / zero rows: zeros, their MFMAs add nothing)
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(r.own ? r.offA : 0u));
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(wgh::A_OFF + BUF * wgh::A_BUF + (wave + 8 * K) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_vptr)(smem + dst), 16, v, soff, 0, 0);
    }
    template <int BUF, int K>
    __device__ __forceinline__ void dma_x(const Row& r) {
        const bool edge = (r.left && (xedge[K] & 1u)) || (r.right && (xedge[K] & 2u));
        const unsigned v = (!r.data || (K == 1 && dummyX1) || edge) ? wgh::kOut : vX[K] - (r.left ? 16u : 0u);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(r.data ? r.offX : 0u));
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((K == 1 && dummyX1) ? wgh::DUMMY_OFF : wgh::XR_OFF + BUF * wgh::XR_BUF + (wave + 8 * K) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_vptr)(smem + dst), 16, v, soff, 0, 0);
    }
    // x raw row (buffer BUF) -> the three column-shifted octets of this thread's (channel, octet, plane) in ring slot SLOT
    unsigned rw[10];
    template <int BUF>
    __device__ __forceinline__ void regroup_read() {
        const char* xr = smem + xr_rd + BUF * wgh::XR_BUF;
        const u32x4 qa = *reinterpret_cast<const u32x4*>(xr + 16);
        const u32x4 qb = *reinterpret_cast<const u32x4*>(xr + 32);
        rw[0] = *reinterpret_cast<const unsigned*>(xr + 12);
        rw[9] = *reinterpret_cast<const unsigned*>(xr + 48);
        rw[1] = qa[0]; rw[2] = qa[1]; rw[3] = qa[2]; rw[4] = qa[3];
        rw[5] = qb[0]; rw[6] = qb[1]; rw[7] = qb[2]; rw[8] = qb[3];
        if constexpr (!XS) {
            // fp32 words: this thread's plane takes bf16(v) (hi) or bf16(v - bf16(v)) (lo) of every value; the values it will pack are
            // prepared once here (the residuals for a lo thread), the three shifts then cost one v_cvt_pk each per pair
#pragma unroll
            for (int k = 0; k < 10; k += 2) {
                unsigned h_, l_;
                split2(__uint_as_float(rw[k]), __uint_as_float(rw[k + 1]), h_, l_);
                // residuals as fp32 again (exact: a bf16 widened): lo thread packs them, hi thread packs the values themselves
                const unsigned r0 = l_ << 16, r1 = l_ & 0xFFFF0000u;
                rw[k] = psel == 0x05040100u ? r0 : rw[k];
                rw[k + 1] = psel == 0x05040100u ? r1 : rw[k + 1];
            }
        }
    }
    template <int SLOT, int S>
    __device__ __forceinline__ void regroup_shift() {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (XS) {
                o[q] = __builtin_amdgcn_perm(rw[S + 2 * q + 1], rw[S + 2 * q], psel);
            } else {       // (hi thread: rounds the values; lo thread: its residuals are bf16 values already -- the conversion is exact)
                const f32x2 v = {__uint_as_float(rw[S + 2 * q]), __uint_as_float(rw[S + 2 * q + 1])};
                o[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            }
        }
        *reinterpret_cast<u32x4*>(smem + pl_wr + SLOT * wgh::PL_SLOT + S * 2 * wgh::PL_BLK) = o;
    }
    // dY fragments of the row in raw buffer BUF -> register set SET: reads, then the hi / lo regrouping in two halves
    u32x4 araw[wgh::RM][2];
    template <int BUF>
    __device__ __forceinline__ void a_read(int i) {
        araw[i][0] = *reinterpret_cast<const u32x4*>(smem + a_q0 + BUF * wgh::A_BUF + i * 2048);
        araw[i][1] = *reinterpret_cast<const u32x4*>(smem + a_q1 + BUF * wgh::A_BUF + i * 2048);
    }
    template <int SET>
    __device__ __forceinline__ void a_regroup(int i) {
        const u32x4 qa = araw[i][0], qb = araw[i][1];
        u32x4 hv, lv;
        if constexpr (AS) {
            hv[0] = __builtin_amdgcn_perm(qa[1], qa[0], 0x07060302u); lv[0] = __builtin_amdgcn_perm(qa[1], qa[0], 0x05040100u);
            hv[1] = __builtin_amdgcn_perm(qa[3], qa[2], 0x07060302u); lv[1] = __builtin_amdgcn_perm(qa[3], qa[2], 0x05040100u);
            hv[2] = __builtin_amdgcn_perm(qb[1], qb[0], 0x07060302u); lv[2] = __builtin_amdgcn_perm(qb[1], qb[0], 0x05040100u);
            hv[3] = __builtin_amdgcn_perm(qb[3], qb[2], 0x07060302u); lv[3] = __builtin_amdgcn_perm(qb[3], qb[2], 0x05040100u);
        } else {
            unsigned h_, l_;
            split2(__uint_as_float(qa[0]), __uint_as_float(qa[1]), h_, l_); hv[0] = h_; lv[0] = l_;
            split2(__uint_as_float(qa[2]), __uint_as_float(qa[3]), h_, l_); hv[1] = h_; lv[1] = l_;
            split2(__uint_as_float(qb[0]), __uint_as_float(qb[1]), h_, l_); hv[2] = h_; lv[2] = l_;
            split2(__uint_as_float(qb[2]), __uint_as_float(qb[3]), h_, l_); hv[3] = h_; lv[3] = l_;
        }
        // (pinned: the fragments are first used in the NEXT step, and the compiler would otherwise sink the v_perms there -- in front of
        // that step's first MFMA instead of into this step's MFMA shadow)
        asm volatile("" : "+v"(hv), "+v"(lv));
        ah[SET][i] = __builtin_bit_cast(bf16x8, hv);
        al[SET][i] = __builtin_bit_cast(bf16x8, lv);
    }
    bf16x8 bh[2], bl[2];            // B fragments of the current / the next tap
    template <int SLOT, int S, int PLANE>
    __device__ __forceinline__ bf16x8 ld_b() {
        return *reinterpret_cast<const bf16x8*>(smem + pl_rd[SLOT] + (2 * S + PLANE) * wgh::PL_BLK);
    }
    __device__ __forceinline__ void shift_rows() {
        rows[0] = rows[1]; rows[1] = rows[2]; rows[2] = rows[3]; rows[3] = rows[4];
        rows[4] = next_row();
    }
    // Step jg, PH = jg mod 6.  Taps T = 3 r + s; tap T's B fragments sit in bh / bl[(PH + T) & 1], fetched during tap T - 1 (tap 0's during
    // the previous step's tap 8).  filler<T, I> is what is issued behind the three MFMAs of (tap T, row block I): the order is pinned with
    // sched_barriers -- left alone the compiler gathers the step's DMA / regroup / fragment work into one block without a single MFMA in it.
    template <int PH, int T, int I>
    __device__ __forceinline__ void filler() {
        constexpr int S_CUR = PH % 3, S_NEXT = (PH + 1) % 3, SET = PH & 1;
        // B fragments of tap T + 1: its row is jg - 1 / jg / jg + 1 for r = 0 / 1 / 2; tap 9 = tap 0 of step jg + 1: row jg
        constexpr int TN = T + 1, RN = (TN % 9) / 3, SN = TN % 3;
        constexpr int SLOTN = (TN == 9) ? S_CUR : (RN == 0 ? (PH + 2) % 3 : (RN == 1 ? S_CUR : S_NEXT));
        if constexpr (T == 0 && I == 2) shift_rows();          // rows[] now describe jg .. jg + 4 (the DMA fillers of taps 3 / 4 read them)
        if constexpr (I == 0) bh[(PH + TN) & 1] = ld_b<SLOTN, SN, 0>();
        if constexpr (I == 1) bl[(PH + TN) & 1] = ld_b<SLOTN, SN, 1>();
        if constexpr (T == 3 && I == 2) dma_x<(PH + 1) % 3, 0>(rows[4]);
        if constexpr (T == 3 && I == 3) dma_x<(PH + 1) % 3, 1>(rows[4]);
        if constexpr (T == 4 && I == 2) dma_a<PH % 3, 0>(rows[3]);
        if constexpr (T == 4 && I == 3) dma_a<PH % 3, 1>(rows[3]);
        if constexpr (T == 5 && I == 2) regroup_read<(PH + 2) % 3>();
        if constexpr (T == 5 && I == 3) { a_read<(PH + 1) % 3>(0); a_read<(PH + 1) % 3>(1); }
        if constexpr (T == 6 && I == 1) a_regroup<SET ^ 1>(0);
        if constexpr (T == 6 && I == 2) regroup_shift<(PH + 2) % 3, 0>();
        if constexpr (T == 6 && I == 3) regroup_shift<(PH + 2) % 3, 1>();
        if constexpr (T == 7 && I == 1) a_regroup<SET ^ 1>(1);
        if constexpr (T == 7 && I == 2) { regroup_shift<(PH + 2) % 3, 2>(); a_read<(PH + 1) % 3>(2); a_read<(PH + 1) % 3>(3); }
        if constexpr (T == 8 && I == 1) a_regroup<SET ^ 1>(2);
        if constexpr (T == 8 && I == 2) a_regroup<SET ^ 1>(3);
    }
    template <int PH, int T, bool MM>
    __device__ __forceinline__ void tap() {
        constexpr int SET = PH & 1;
#pragma unroll
        for (int i = 0; i < wgh::RM; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MM) {
                constexpr int BI = (PH + T) & 1;       // (9 taps per step: the buffer parity of tap T alternates from step to step)
                acc[i][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[SET][i], bh[BI], acc[i][T], 0, 0, 0);
                acc[i][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[SET][i], bl[BI], acc[i][T], 0, 0, 0);
                acc[i][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[SET][i], bh[BI], acc[i][T], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) filler<PH, T, 0>();
            if (i == 1) filler<PH, T, 1>();
            if (i == 2) filler<PH, T, 2>();
            if (i == 3) filler<PH, T, 3>();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int PH, bool MM>
    __device__ __forceinline__ void step() {
        tap<PH, 0, MM>(); tap<PH, 1, MM>(); tap<PH, 2, MM>();
#ifdef AG2V_WGH_LOOSE_WAIT       // TIMING EXPERIMENT ONLY (races): is the counted wait -- DMA latency -- what the waves park on?
        wgh_wait_vmcnt<4 + AG2V_WGH_LOOSE_WAIT>();
#else
        wgh_wait_vmcnt<4>();                                  // the 4 pieces of step jg - 1 may stay in flight
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's plane stores of step jg - 1
        __builtin_amdgcn_s_barrier();
        tap<PH, 3, MM>(); tap<PH, 4, MM>(); tap<PH, 5, MM>();
        tap<PH, 6, MM>(); tap<PH, 7, MM>(); tap<PH, 8, MM>();
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        ars = make_rsrc(p.a_src, p.a_bytes);
        brs = make_rsrc(p.b_src, p.b_bytes);
        H = p.H; W = p.W; HW = H * W; M = p.M; C = p.C;
        spr = W / wgh::TW;
        // (split, tile): consecutive ids on one XCD are the tiles of one split -- they read the same dY / x rows (see the v kernels)
        const int ntiles = p.tiles_m * p.tiles_n;
        const int gl = xcd_remap(blockIdx.x, ntiles * p.splitk);
        const int split = gl / ntiles, lid = gl - split * ntiles;
        const int tile_m = lid % p.tiles_m, tile_c = lid / p.tiles_m;
        m0 = tile_m * wgh::BM;
        c0 = tile_c * wgh::CB;
        const int grp = split / p.spg, sl = split - grp * p.spg;
        nst_g = (p.gq / HW) * spr;                             // strips of a sample group
        g0 = sl * p.kchunk;                                    // (kchunk: ring rows per split)
        g1 = min(g0 + p.kchunk, nst_g * (H + 1));
        float* const outg = p.det ? p.out + (size_t)split * ((size_t)p.M * p.NG) : p.out + (size_t)grp * p.out_gstride;
        const bool atomic_out = !p.det && (p.spg > 1 || p.accumulate);
        if (g1 <= g0) return;
        // ---- DMA source offsets
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pa = wave + 8 * k;
            const int row = 8 * pa + (lane >> 3), qs = lane & 7, q = qs ^ wgh_aswz(row);
            vA[k] = (m0 + row < M) ? (unsigned)(m0 + row) * (unsigned)HW * 4u + (unsigned)q * 16u : wgh::kOut;
            const int Q = 64 * pa + lane, c = Q / 10, qq = Q - 10 * c;
            vX[k] = (unsigned)c * (unsigned)HW * 4u + (unsigned)qq * 16u;
            xedge[k] = (qq == 0 ? 1u : 0u) | (qq == 9 ? 2u : 0u) | (Q >= 640 ? 3u : 0u);
        }
        dummyX1 = wave + 8 >= 10;
        if (dummyX1) { xedge[1] = 3u; vX[1] = 0u; }
        // ---- fragment / regroup addresses
        const int wm = wave >> 2, wn = wave & 3;
        const int ml = lane & 15, g = lane >> 4;
        const int arow = wm * 64 + ml, fsw = wgh_aswz(arow);
        a_q0 = (unsigned)(wgh::A_OFF + arow * 128 + (((2 * g) ^ fsw) * 16));
        a_q1 = (unsigned)(wgh::A_OFF + arow * 128 + (((2 * g + 1) ^ fsw) * 16));
        const int cc = wn * 16 + ml;
        const int sig_r = (0x78 >> (2 * ((cc >> 2) & 3))) & 3;         // sigma = [0, 2, 3, 1][(c >> 2) & 3]: see the bank notes in DESIGN.md
#pragma unroll
        for (int k = 0; k < 3; ++k) pl_rd[k] = (unsigned)(wgh::PL_OFF + k * wgh::PL_SLOT + cc * 64 + ((g ^ sig_r) * 16));
        const int uc = tid >> 3, uo = (tid >> 1) & 3, up = tid & 1;
        const int sig_w = (0x78 >> (2 * ((uc >> 2) & 3))) & 3;
        xr_rd = (unsigned)(wgh::XR_OFF + uc * wgh::XR_ROW + uo * 32);
        pl_wr = (unsigned)(wgh::PL_OFF + up * wgh::PL_BLK + uc * 64 + ((uo ^ sig_w) * 16));
        psel = up ? 0x05040100u : 0x07060302u;
        asm volatile("" : "+v"(a_q0), "+v"(a_q1), "+v"(pl_rd[0]), "+v"(pl_rd[1]), "+v"(pl_rd[2]), "+v"(xr_rd), "+v"(pl_wr));
        // ---- row stream: the cursor starts at ring row g0 - 1 (its x is the first owned row's upper tap row; -1: in front of the group)
        {
            cr = g0 - 1;
            const int rr = cr < 0 ? 0 : cr;
            const int sabs = rr / (H + 1);                   // strip number inside the group
            cst = sabs;
            cy = rr - sabs * (H + 1) - 1;
            cn = grp * (p.gq / HW) + sabs / spr;
            cxs = sabs % spr;
        }
        // rows jg = J0 - 5 .. J0 - 1 do not exist: five zero rows in front of the cursor
        Row zero = {0, 0u, 0u, 0, 0};
        rows[0] = zero;                                       // jg = J0 - 5
        rows[1] = zero; rows[2] = zero; rows[3] = zero; rows[4] = zero;
        // (every step shifts the window by one row behind its first MFMAs: the first fill step's window ends at J0 - 1)
#pragma unroll
        for (int i = 0; i < wgh::RM; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.f;
        // pipeline fill: steps J0 - 5 .. J0 - 1 (phases 1 .. 5, so that J0 is phase 0)
        step<1, false>(); step<2, false>(); step<3, false>(); step<4, false>(); step<5, false>();
        // rows[0] is now ring row g0: one step per owned row
        const int total = g1 - g0;
        int done = 0;
        while (true) {
            step<0, true>(); if (++done == total) break;
            step<1, true>(); if (++done == total) break;
            step<2, true>(); if (++done == total) break;
            step<3, true>(); if (++done == total) break;
            step<4, true>(); if (++done == total) break;
            step<5, true>(); if (++done == total) break;
        }
        wgh_wait_vmcnt<0>();          // (the look-ahead DMAs of the last steps -- all out of range -- still target this workgroup's LDS)
        // ---- epilogue: tile (i, t) register r: dY channel m0 + wm * 64 + 16 i + 4 g + r, x channel c0 + wn * 16 + ml, tap t
        const int col = (c0 + cc) * 9;
#pragma unroll
        for (int i = 0; i < wgh::RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + 16 * i + 4 * g + r;
                if (m < M) {
                    float* dst = outg + (size_t)m * p.NG + col;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        if (atomic_out) atomicAdd(dst + t, acc[i][t][r]);
                        else dst[t] = acc[i][t][r];
                    }
                }
            }
    }
};
#endif

template <bool AS, bool XS>
__global__ __launch_bounds__(512) void conv_wgrad_halo_bf16x3_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) char wgh_smem[wgh::LDS_BYTES];
    WgradHaloBody<AS, XS> b(p, wgh_smem);
    b.run();
#endif
}

// ---------------------------------------------------------------------------------------------------------------- host side
bool wgrad_halo_switched_off();       // conv_halo.hip: AG2V_NO_WGRAD_HALO, read with the other A/B switches (ag2v_halo_reload_env)
bool wgrad_halo_ok(const ConvArgs& a) {
    if (wgrad_halo_switched_off() || a.f32) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.OH != a.H || a.OW != a.W) return false;
    if (a.W % wgh::TW || a.H < 16 || a.C % wgh::CB || a.M < 48) return false;       // (a 64-row layer fills half of a row tile: still ahead of the 16-byte-load kernels, which pad the same way)
    if (a.groups < 1 || a.gq % (a.H * a.W)) return false;
    if (a.a_bytes == 0 || a.b_bytes == 0) return false;
    return true;
}
// tiles, strips per split (kchunk), splits per group
void plan_wgrad_halo(ConvArgs& a) {
    a.tiles_m = (a.M + wgh::BM - 1) / wgh::BM;
    a.tiles_n = a.C / wgh::CB;
    const int tiles = a.tiles_m * a.tiles_n;
    if (a.groups < 1) { a.groups = 1; a.gq = a.KG; a.out_gstride = (long long)a.M * a.NG; }
    // ring rows of a sample group: its strips' rows, each strip behind one zero row; any split of them into equal runs is a valid
    // partition (a run may start and end inside a strip), so every layer can fill all 256 CUs
    const long long rows_g = (long long)(a.gq / (a.H * a.W)) * (a.W / wgh::TW) * (a.H + 1);
    // (Workgroups that read the same dY / x rows -- the tiles of one split -- only find each other's lines in their XCD's L2 while they
    // walk in step, and over thousands of steps they do not: FETCH_SIZE x3.6 of the algorithmic bytes on the merged mlp_shared shape,
    // x1.03 on the 4-tile gamma | beta shape.  Shorter workgroups, AG2V_WGH_MAX_STEPS, do not restore it and cost 0.6-3 %.)
    static int max_steps = 0;
    if (max_steps == 0) {
        max_steps = 1 << 30;
    }
    int best = 1;
    long long best_cost = -1;
    for (int spg = 1; spg <= 4096 && spg <= rows_g; ++spg) {
        const long long chunk = (rows_g + spg - 1) / spg;
        if ((rows_g + chunk - 1) / chunk != spg) continue;
        if (spg < 4096 && chunk > max_steps) continue;
        const long long wgs = (long long)tiles * a.groups * spg;
        const long long rounds = (wgs + 255) / 256;                 // one workgroup per CU (LDS)
        const long long cost = rounds * (chunk + 8);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = spg; }
    }
    a.spg = best;
    a.kchunk = (int)((rows_g + best - 1) / best);
    a.splitk = best * a.groups;
}
void set_last_conv_kernel(const char* name);
int launch_wgrad_halo(ConvArgs& a, hipStream_t st) {
    plan_wgrad_halo(a);
    if (a.spg > 1 && !a.accumulate && !a.det) {
        if (hipMemsetAsync(a.out, 0, (size_t)a.M * a.NG * a.groups * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    }
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * a.splitk));
    if (a.a_split && a.b_split) {
        set_last_conv_kernel("conv_wgrad_halo_bf16x3_kernel<true, true>");
        hipLaunchKernelGGL((conv_wgrad_halo_bf16x3_kernel<true, true>), grid, dim3(512), 0, st, a);
    } else if (a.a_split) {
        set_last_conv_kernel("conv_wgrad_halo_bf16x3_kernel<true, false>");
        hipLaunchKernelGGL((conv_wgrad_halo_bf16x3_kernel<true, false>), grid, dim3(512), 0, st, a);
    } else if (a.b_split) {
        set_last_conv_kernel("conv_wgrad_halo_bf16x3_kernel<false, true>");
        hipLaunchKernelGGL((conv_wgrad_halo_bf16x3_kernel<false, true>), grid, dim3(512), 0, st, a);
    } else {
        set_last_conv_kernel("conv_wgrad_halo_bf16x3_kernel<false, false>");
        hipLaunchKernelGGL((conv_wgrad_halo_bf16x3_kernel<false, false>), grid, dim3(512), 0, st, a);
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
