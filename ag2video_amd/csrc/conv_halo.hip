// 3x3 stride-1 convolution on the bf16 matrix pipe ("bf16x3", see conv_bf16x3.hip) with the gathered operand staged as HALO TILES by
// LDS-DMA (buffer_load ... lds): no producer waves, no VALU and no ds_write on the way from memory to LDS.  gfx950 only.
//
// Replaces, for the shapes it covers, conv_igemm_bf16x3_ps_kernel: forward and stride-1 data gradient of every 3x3 convolution whose
// gathered tensor holds SPLIT WORDS ((bf16 hi << 16) | bf16 lo per value: the SPADE decoder's activations and gradients, reference
// models/spade_models/networks/architecture.py:29-32 conv_0 / conv_1, normalization.py:90-94 mlp_shared / mlp_gamma / mlp_beta).
//
// Why: in the wave-specialised kernels 8 producer waves spend a K-tile issuing 4..16-byte gathers (every filter tap re-fetches the same
// pixels shifted by one), regrouping hi / lo halves and storing them to LDS, next to 8 consumer waves -- each side alone needs ~2 000
// cycles per K-tile, together ~2 700.  Here a workgroup (8 waves, all of them multiply) owns an output patch of TH rows
// x TW columns and stages, per 16 input channels, ONE image of the (TH + 2) x (TW + 8) input pixels around it: all nine taps read it
// at different offsets, so memory -> LDS traffic drops ~4x against tap-by-tap gathers and the copy is 1 KiB DMA pieces
// (lane-linear LDS destination, per-lane source: 64 quads of 4 pixels) whose per-lane offsets are computed once per workgroup -- a
// stage only moves a wave-uniform SGPR offset.  Padding costs nothing: out-of-image quads load from an out-of-range buffer offset, for
// which the DMA writes zeros to the lane's 16 LDS bytes (tools/ubench/lds_dma_probe.hip), so every stage re-creates its own zero border.
// Workgroups are PERSISTENT (one per CU, which is all the LDS footprint admits): the tail of a tile's pipeline -- which would otherwise
// fetch nothing -- already loads the first image and weight steps of the workgroup's next tile, so only the first tile of a launch
// exposes its DMA latency and a tile's epilogue stores run under the next tile's loads.
// The weights arrive as the bf16 hi / lo planes the tap-major pack kernels already write ([KG/8][M][8], conv_igemm.hip
// pack_fwd_tap_kernel / pack_dgrad_tap_kernel) and go to a 6-slot LDS ring the same way, one 16-deep k-step (16 channels of one tap)
// per slot, three steps ahead.
//
// A wave owns EM x 32 output channels x 32 pixels.  Per k-step it reads its B fragment as 8 split words (ds_read_b32 at immediate
// offsets: channel j, tap (r, s)), regroups them into the hi / lo bf16x8 operands (8 v_perm) and issues 3 EM MFMAs; the operands of
// step t + 1 are requested during step t (B from the staged image, A from the ring slot certified by the barrier at the top of
// step t), so the matrix pipe starts right behind each barrier.  One s_barrier and one counted s_waitcnt vmcnt per step.
#include <type_traits>
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_vptr;

#ifndef AG2V_HALO_NA
#define AG2V_HALO_NA 6          // A ring slots (divides the 18-step unrolled body)
#endif

template <int WM, int WN, int EM, int TW, int TH, bool MF16 = false, int CB = 2>
struct HaloCfg {
    // CB: 16-pixel column blocks per wave (MF16 only).  2: the wave owns 32 pixels; 4 (round 6, the 64-row tile): 64 pixels -- a 64-row
    // wave with 32 pixels reads as many B words per MFMA as a 128-row wave reads A and B together and is bound by them (matrix pipe
    // 47 % busy); with 64 pixels per wave the 64-row tile has the 128-row tile's operand traffic per MFMA
    static constexpr int BM = WM * EM * 32, BN = WN * 16 * CB;
    static_assert(CB == 2 || (CB == 4 && MF16 && TW % 64 == 0), "column blocks per wave");
    static_assert(WM * WN == 8, "8 waves");
    static_assert(TW * TH == BN && TW % 32 == 0, "pixel patch");
    static constexpr int CH = 16;                         // channels per stage = one MFMA k-step
    static constexpr int RP = TW + 8, NR = TH + 2;        // staged row pitch (words) / rows: one quad of halo columns each side
    static constexpr int CHP = NR * RP;                   // words per channel
    // MF16 (16x16x32 MFMAs): lanes 0-15 / 16-31 of a B-fragment read belong to the two 8-channel groups of the stage; a 16-word skew
    // between the groups puts them on disjoint banks (8 * CHP is a multiple of 32 words for every patch geometry)
    static constexpr int SKEW = MF16 ? 16 : 0;            // words between channel group 0 and channel group 1
    static constexpr int G8 = 8 * CHP + SKEW;             // words from group 0 to group 1
    static constexpr int BQ = (CH * CHP + SKEW) / 4;      // quads per stage
    static constexpr int NPB = (BQ + 63) / 64;            // DMA pieces (64 quads = 1 KiB) per stage
    static constexpr int PPW = (NPB + 7) / 8;             // ... per wave
    static constexpr int BBYTES = NPB * 1024;
    static constexpr int ASZ = BM * 64;                   // bytes per A step: [plane hi | lo][k group 0 | 1][BM rows][8 bf16]
    static constexpr int APS = ASZ / 1024;                // A pieces per step
    static constexpr int APW = (APS + 7) / 8;
    static constexpr int NA = AG2V_HALO_NA;
    // (16x16 form: the launch's bias vector, zero-padded to whole row tiles, behind the dummy piece -- at most BIAS_ROWS output channels)
    static constexpr int BIAS_ROWS = MF16 ? 2048 + 128 : 0;
    // ... and, behind it, the per-row sums of what the workgroup's epilogues wrote (ConvArgs::chan_sums)
    static constexpr int A_OFF = 0, B_OFF = NA * ASZ, DUMMY_OFF = B_OFF + 2 * BBYTES, BIAS_OFF = DUMMY_OFF + 1024, SUM_OFF = BIAS_OFF + BIAS_ROWS * 4,
                         LDS_BYTES = SUM_OFF + BIAS_ROWS * 4;
    static_assert(PPW <= (MF16 ? 7 : 6), "a stage's pieces must be certified before the next stage's first prefetch");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((7 * CHP + 2 * RP + 2) * 4 < 65536 && (2 * ASZ + 3 * BM * 16 + EM * 512) < 65536, "ds_read immediate offsets");
};

#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the kernel stubs; it rejects some of the device-only template code)
// (measured and NOT used: the two waves of a SIMD issuing their DMA pieces behind different MFMA groups -- -1..-3 % on every shape)
#ifdef AG2V_HALO_STAGGER
__device__ __forceinline__ bool getenv_stagger() { return true; }
#else
__device__ __forceinline__ bool getenv_stagger() { return false; }
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WM, int WN, int EM, int TW, int TH, bool PS>
struct HaloBody {
    typedef HaloCfg<WM, WN, EM, TW, TH> G;
    static constexpr unsigned kOut = 0xFFFFFFF0u;       // an out-of-range buffer offset: the DMA writes zeros for that lane
    const ConvArgs& p;
    char* smem;
    __amdgpu_buffer_rsrc_t ars, brs;
    int wave, lane;
    // DMA state: per-lane source offsets of this wave's pieces for the CURRENT tile (vA, vB) and for the workgroup's NEXT tile (vAn,
    // vBn: the tail of a tile's pipeline already fetches the next tile's first image and weight steps)
    unsigned vA[G::APW], vB[G::PPW], vAn[G::APW], vBn[G::PPW];
    unsigned dstB0[G::PPW];              // LDS byte offset of this wave's B pieces inside a stage buffer (or the dummy KiB, absolute)
    bool dummyB[G::PPW];
    unsigned dstA[G::APW];               // LDS byte offset inside an A slot (or the dummy KiB, absolute)
    bool dummyA[G::APW];
    unsigned hw4;                        // H * W * 4
    int NS, lastP, P;                    // stages, last stage pair, current pair
    // fragment state
    unsigned baseA[2], baseB[2];
    bf16x8 ah[2][EM], al[2][EM], bh[2], bl[2];
    unsigned bw[8];
    f32x16 acc[EM];

    __device__ __forceinline__ HaloBody(const ConvArgs& p_, char* s) : p(p_), smem(s) {}

    template <int U3>
    __device__ __forceinline__ void dma_a(int Pn, const unsigned (&v)[G::APW]) {      // the A step U3 steps into pair Pn (U3 < 18)
        constexpr int slot = U3 % G::NA;
        constexpr int c = (U3 % 9) * 4 + 2 * (U3 / 9);       // k-group row of the step inside its 32-channel chunk: (tap, half)
        const unsigned soff = (unsigned)(Pn * 36 + c) * (unsigned)p.M * 16u;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            const unsigned dst = dummyA[j] ? dstA[j] : (unsigned)(G::A_OFF + slot * G::ASZ) + dstA[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_vptr)(smem + dst), 16, v[j], soff, 0, 0);
        }
    }
    template <int K, int BUF>
    __device__ __forceinline__ void dma_b(unsigned voff, unsigned soff) {       // piece round K into stage buffer BUF
        const unsigned dst = dummyB[K] ? dstB0[K] : (unsigned)(G::B_OFF + BUF * G::BBYTES) + dstB0[K];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_vptr)(smem + dst), 16, voff, soff, 0, 0);
    }
    // Operand registers: set (U & 1) holds step U's A fragments (ah / al) and its B operand already regrouped (bh / bl).
    template <int U1>
    __device__ __forceinline__ void load_a(int i) {          // A fragments of row block i for the step U1 steps into the pair
        constexpr int set = U1 & 1, slot = U1 % G::NA;
        const char* ab = smem + baseA[slot / 3] + (slot % 3) * G::ASZ;
        ah[set][i] = *reinterpret_cast<const bf16x8*>(ab + i * 512);
        al[set][i] = *reinterpret_cast<const bf16x8*>(ab + 2 * G::BM * 16 + i * 512);
    }
    template <int U1>
    __device__ __forceinline__ unsigned load_bw(int j) {     // word of channel j (of this lane's k group) at step U1's tap
        constexpr int buf = (U1 / 9) & 1, tap = U1 % 9, r = tap / 3, s = tap % 3;
        return *reinterpret_cast<const unsigned*>(smem + baseB[buf] + (r * G::RP + s) * 4 + j * G::CHP * 4);
    }
    template <int U1>
    __device__ __forceinline__ void regroup() {              // 8 words -> hi / lo bf16x8 operands of set U1 & 1
        constexpr int set = U1 & 1;
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        u32x4v hv, lv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (PS) {          // split words: regroup the halves
                hv[q] = __builtin_amdgcn_perm(bw[2 * q + 1], bw[2 * q], 0x07060302u);
                lv[q] = __builtin_amdgcn_perm(bw[2 * q + 1], bw[2 * q], 0x05040100u);
            } else {                     // fp32 words: the 3-term split here, in the wave that multiplies (6 VALU per pair)
                unsigned h_, l_;
                split2(__uint_as_float(bw[2 * q]), __uint_as_float(bw[2 * q + 1]), h_, l_);
                hv[q] = h_;
                lv[q] = l_;
            }
        }
        bh[set] = __builtin_bit_cast(bf16x8, hv);
        bl[set] = __builtin_bit_cast(bf16x8, lv);
    }
    template <int U1>
    __device__ __forceinline__ void prefetch_all() {         // (tile start) every operand of step U1 at once
#pragma unroll
        for (int i = 0; i < EM; ++i) load_a<U1>(i);
#pragma unroll
        for (int j = 0; j < 8; ++j) bw[j] = load_bw<U1>(j);
        regroup<U1>();
    }
    template <int U>
    __device__ __forceinline__ void step() {
        constexpr int tap = U % 9, ptap = (U + 8) % 9, set = U & 1;
        // everything this wave issued before the previous step has landed; the barrier makes that true for every wave's pieces: the A
        // slots of steps t, t + 1 and (from tap PPW + 1 on) the next stage's image are certified
        wait_vmcnt<G::APW + (ptap < G::PPW ? 1 : 0)>();
        __builtin_amdgcn_s_barrier();
        // EM groups of {3 MFMAs on this step's operands (in registers since the previous step), the LDS reads of 1 / EM of the next
        // step's operands}.  The order is pinned: left alone the scheduler sinks every read to just in front of its use (fewest live
        // registers), which puts an LDS round trip in front of every other MFMA.  The step's DMA pieces are issued BEHIND the first
        // MFMA group: an LDS-DMA instruction takes 60-180 cycles to issue (MI355X_MICROARCH.md constants) and both waves of a SIMD issue
        // theirs at the same point of the step -- at the top of the step the matrix pipe would idle for that long after every barrier
        // (128-row tile: 446 -> 464 TF).
        constexpr int WPG = 8 / EM;
#pragma unroll
        for (int g = 0; g < EM; ++g) {
            __builtin_amdgcn_sched_barrier(0);
#ifndef AG2V_HALO_FAKE16
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][g], bh[set], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][g], bl[set], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][g], bh[set], acc[g], 0, 0, 0);
#else
            // TIMING EXPERIMENT ONLY (wrong results): the same operands, matrix-pipe cycles and accumulator registers through the 16x16x32
            // shape -- does the chip hold a higher clock on it in this loop (MI355X_MICROARCH.md, DVFS give-back item 7)?
            {
                f32x4 q0 = __builtin_shufflevector(acc[g], acc[g], 0, 1, 2, 3), q1 = __builtin_shufflevector(acc[g], acc[g], 4, 5, 6, 7);
                f32x4 q2 = __builtin_shufflevector(acc[g], acc[g], 8, 9, 10, 11), q3 = __builtin_shufflevector(acc[g], acc[g], 12, 13, 14, 15);
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[set][g], bh[set], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[set][g], bh[set], q1, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][g], bl[set], q2, 0, 0, 0);
                q3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][g], bl[set], q3, 0, 0, 0);
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][g], bh[set], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[set][g], bh[set], q1, 0, 0, 0);
                acc[g] = __builtin_shufflevector(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                 __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {
                // beyond this tile's last stage / step the pipeline runs on into the NEXT tile of this workgroup (its stage 0 goes to
                // buffer 0, its steps 0..2 to slots 0..2 -- free by then); without a next tile the offsets are out of range (zeros)
                if constexpr (tap < G::PPW) {
                    const int st = 2 * P + U / 9 + 1;
                    const bool nx = st >= NS;
                    dma_b<tap, ((U / 9) + 1) & 1>(nx ? vBn[tap] : vB[tap], nx ? 0u : (unsigned)st * 16u * hw4);
                }
                if constexpr (U + 3 < 18) dma_a<U + 3>(P, vA);
                else if (P < lastP) dma_a<U + 3 - 18>(P + 1, vA);
                else dma_a<U + 3 - 18>(0, vAn);
                __builtin_amdgcn_sched_barrier(0);
            }
            load_a<U + 1>(g);
#pragma unroll
            for (int j = 0; j < WPG; ++j) bw[g * WPG + j] = load_bw<U + 1>(g * WPG + j);
        }
        __builtin_amdgcn_sched_barrier(0);
        regroup<U + 1>();
        __builtin_amdgcn_sched_barrier(0);
    }

    struct Tile { int n_img, y0, x0, m0; };
    // source offsets of this wave's DMA pieces for logical tile v (v < 0: none -- every lane out of range)
    __device__ __forceinline__ Tile describe(int v, unsigned (&va)[G::APW], unsigned (&vb)[G::PPW]) {
        Tile t = {0, 0, 0, 0};
        const bool live = v >= 0;
        const int lid = live ? xcd_remap(v, p.tiles_m * p.tiles_n) : 0;
        const int tile_m = lid % p.tiles_m, tn = lid / p.tiles_m;
        const int tcols = p.OW / TW, tpi = (p.OH / TH) * tcols;
        t.n_img = tn / tpi;
        const int trem = tn - t.n_img * tpi;
        t.y0 = (trem / tcols) * TH;
        t.x0 = (trem % tcols) * TW;
        t.m0 = tile_m * G::BM;
        const unsigned lo_plane = (unsigned)(p.KG >> 3) * (unsigned)p.M * 16u;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            const int e = (wave + 8 * j) * 64 + lane;
            const int plane = e / (2 * G::BM), kg = (e / G::BM) & 1, m = e % G::BM;
            va[j] = (live && !dummyA[j] && t.m0 + m < p.M) ? (unsigned)plane * lo_plane + (unsigned)(kg * p.M + t.m0 + m) * 16u : kOut;
        }
#pragma unroll
        for (int k = 0; k < G::PPW; ++k) {
            const int q = (wave + 8 * k) * 64 + lane;
            const int ch = q / (G::CHP / 4), w = q % (G::CHP / 4);
            const int row = w / (G::RP / 4), cq = w % (G::RP / 4);
            const int y = t.y0 - 1 + row, x = t.x0 - 4 + 4 * cq;
            const bool ok = live && !dummyB[k] && q < G::BQ && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            vb[k] = ok ? (unsigned)(((t.n_img * p.C + ch) * p.H + y) * p.W + x) * 4u : kOut;
        }
        return t;
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        ars = make_rsrc(p.a_src, p.a_bytes);
        brs = make_rsrc(p.b_src, p.b_bytes);
        hw4 = (unsigned)(p.H * p.W) * 4u;
        NS = p.C / 16;
        lastP = NS / 2 - 1;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            dummyA[j] = wave + 8 * j >= G::APS;
            dstA[j] = dummyA[j] ? (unsigned)G::DUMMY_OFF : (unsigned)(wave + 8 * j) * 1024u;
        }
#pragma unroll
        for (int k = 0; k < G::PPW; ++k) {
            dummyB[k] = wave + 8 * k >= G::NPB;
            dstB0[k] = dummyB[k] ? (unsigned)G::DUMMY_OFF : (unsigned)(wave + 8 * k) * 1024u;
        }
        // ---- fragment addresses (the same for every tile)
        const int wm = wave / WN, wn = wave % WN;
        const int wm0 = wm * EM * 32;
        const int pb = wn * 32, ty = pb / TW, tx = pb % TW;
        const int kg = lane >> 5, l31 = lane & 31;
        baseA[0] = (unsigned)(G::A_OFF + kg * G::BM * 16 + (wm0 + l31) * 16);
        baseA[1] = baseA[0] + 3u * G::ASZ;
        baseB[0] = (unsigned)(G::B_OFF + ((kg * 8) * G::CHP + ty * G::RP + tx + l31 + 3) * 4);
        baseB[1] = baseB[0] + (unsigned)G::BBYTES;
        // opaque to the optimiser: it would fold the region offsets (up to 150 KB) into the per-read constants, which then no longer fit
        // the 16-bit ds_read offset field and become one address register per (channel, tap) -- dozens of them, spilled
        asm volatile("" : "+v"(baseA[0]), "+v"(baseA[1]), "+v"(baseB[0]), "+v"(baseB[1]));
        // ---- persistent tile loop: workgroup b takes logical tiles b, b + grid, ... (the grid is a multiple of 8, so they stay on the
        // XCD whose L2 holds the neighbouring tiles: xcd_remap).  The first tile's stage 0 / steps 0..2 are the only exposed DMA latency.
        const int ntiles = p.tiles_m * p.tiles_n, stride = gridDim.x;
        int v = blockIdx.x;
        Tile nxt = describe(v, vAn, vBn);
        P = lastP;                       // (dma_a's "next tile" form)
        if constexpr (G::PPW > 0) dma_b<0, 0>(vBn[0], 0u);
        if constexpr (G::PPW > 1) dma_b<1, 0>(vBn[1 % G::PPW], 0u);
        if constexpr (G::PPW > 2) dma_b<2, 0>(vBn[2 % G::PPW], 0u);
        if constexpr (G::PPW > 3) dma_b<3, 0>(vBn[3 % G::PPW], 0u);
        if constexpr (G::PPW > 4) dma_b<4, 0>(vBn[4 % G::PPW], 0u);
        if constexpr (G::PPW > 5) dma_b<5, 0>(vBn[5 % G::PPW], 0u);
        dma_a<0>(0, vAn);
        dma_a<1>(0, vAn);
        dma_a<2>(0, vAn);
        const int plane = p.OH * p.OW;
        const float* aux0 = p.residual ? p.residual : p.mask;           // at most one of them (the entry points reject both)
        const float ninf = -__builtin_huge_valf();
        const float rmul = p.residual ? 1.f : 0.f;
        const float relu_floor = p.relu ? 0.f : ninf;
        const float mask_thr = (!p.residual && p.mask) ? 0.f : ninf;
        const bool osplit = p.out_split != 0;
        while (true) {
            const Tile cur = nxt;
#pragma unroll
            for (int j = 0; j < G::APW; ++j) vA[j] = vAn[j];
#pragma unroll
            for (int k = 0; k < G::PPW; ++k) vB[k] = vBn[k];
            v += stride;
            const bool more = v < ntiles;
            nxt = describe(more ? v : -1, vAn, vBn);
#pragma unroll
            for (int i = 0; i < EM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            wait_vmcnt<0>();             // this tile's stage 0 and steps 0..2 (issued by the previous tile's tail, or just above)
            __builtin_amdgcn_s_barrier();
            prefetch_all<0>();
            for (P = 0; P <= lastP; ++P) {
                step<0>(); step<1>(); step<2>(); step<3>(); step<4>(); step<5>(); step<6>(); step<7>(); step<8>();
                step<9>(); step<10>(); step<11>(); step<12>(); step<13>(); step<14>(); step<15>(); step<16>(); step<17>();
            }
            // ---- epilogue (the next tile's first pieces are landing meanwhile).  Accumulator register r of block i holds row
            // (r & 3) + 8 (r >> 2) + 4 kg, pixel l31 of the wave's 32.
            const size_t pxo = (size_t)(cur.y0 + ty) * p.OW + (cur.x0 + tx + l31);
            const size_t pix = (size_t)cur.n_img * p.M * plane + pxo;
            float* obase = p.out + (size_t)cur.n_img * p.out_nstride + pxo;
            // (multi-output launch: this row tile's own tensor; rows are addressed by their absolute index below)
            if (p.out_mstride) obase += (long long)(cur.m0 / G::BM) * p.out_mstride - (long long)cur.m0 * plane;
            const float* aux = aux0 ? aux0 + pix : nullptr;
            const float gs = (p.gscale ? p.gscale[cur.n_img / p.group_n] : 1.f) * (p.oscale != 0.f ? p.oscale : 1.f);
#pragma unroll
            for (int i = 0; i < EM; ++i) {
                const int mblk = cur.m0 + wm0 + i * 32;
                float bvec = 0.f;
                if (p.bias) {
                    const int mb = mblk + l31;
                    bvec = (mb < p.M) ? p.bias[mb] : 0.f;
                }
                float ax[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[r] = 0.f;
                if (aux) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mblk + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        if (m < p.M) ax[r] = aux[(size_t)m * plane];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const float b0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bvec), rr));
                    const float b1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bvec), rr + 4));
                    float x = fmaf(acc[i][r], gs, kg ? b1 : b0);
                    x = fmaf(ax[r], rmul, x);
                    x = fmaxf(x, relu_floor);
                    x = (ax[r] > mask_thr) ? x : 0.f;
                    if (osplit) x = pack_split(x);
                    const int m = mblk + rr + 4 * kg;
                    if (m < p.M) obase[(size_t)m * plane] = x;
                }
            }
            if (!more) break;
        }
        wait_vmcnt<0>();          // (the tail DMAs of the last tile -- all out of range -- still target this workgroup's LDS)
    }
};


// ---- the same pipeline on v_mfma_f32_16x16x32_bf16.  The chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back
// item 7; a timing-only build of the 32x32x16 kernel with its MFMAs replaced one-for-two ran +5 % (128-row tile) / +10 % (256-row tile)).
// One instruction spans K = 32 = TWO k-steps of the 32x32x16 pipeline: lanes 0-31 (k groups 0, 1) take the 16 channels of step 2D, lanes
// 32-63 (k groups 2, 3) those of step 2D + 1 -- another tap of the same staged image (or tap 8 of one stage with tap 0 of the next: 18
// steps per stage pair = 9 double-steps), whose ring slot / image offset differs from the first one's by a compile-time constant that sits
// in the lane's base address.  One barrier, one counted wait per DOUBLE-step (48 MFMAs per wave at EM = 4).
// H16 (with PS = false): the fp16 3-term split of the forward passes that need fp32-class products (ConvArgs::h16: the flow network,
// conv_common.h split2_h) -- v_mfma_f32_16x16x32_f16 on planes the *_h pack kernels wrote; the epilogue's oscale undoes their scale
template <int WM, int WN, int EM, int TW, int TH, bool PS, bool H16 = false, int CB = 2>
struct HaloBody16 {
    static_assert(!(PS && H16), "split words are bf16 pairs");
    typedef HaloCfg<WM, WN, EM, TW, TH, true, CB> G;
    static constexpr unsigned kOut = 0xFFFFFFF0u;
    static constexpr int RB = 2 * EM;                    // 16-row blocks of a wave's EM * 32 rows
    // HALFWIN (EM = 4): the A fragments of a double-step are fetched in two halves -- row blocks 0 .. RB/2 - 1 during the previous
    // double-step, RB/2 .. RB - 1 during the double-step itself (5 instead of 8 row blocks live: the full window pushed loop invariants
    // into scratch).  The ring slots of double-step D are then still being read during D, so the DMA pieces that refill them are issued
    // by row halves: the rows of the early blocks three double-steps ahead (as before), the rows of the late blocks only TWO ahead, i.e.
    // after the barrier that ends their last readers' double-step.  A wave's pieces all lie in one row half (lateA).
    static constexpr bool HALFWIN = EM == 4;
    const ConvArgs& p;
    char* smem;
    __amdgpu_buffer_rsrc_t ars, brs;
    int wave, lane;
    unsigned vA[G::APW], vB[G::PPW], vAn[G::APW], vBn[G::PPW];
    unsigned dstB0[G::PPW], dstA[G::APW];
    bool dummyB[G::PPW], dummyA[G::APW];
    unsigned hw4;
    int NS, lastP, P;
    bool lateA;                          // this wave's A pieces hold rows of the late row blocks (HALFWIN)
    int dma_g;                           // MFMA group behind which this wave issues its DMA pieces (0, or RB / 2 for waves 4..7)
    unsigned baseA[3];                   // per slot pair (2 q, 2 q + 1): lanes of the second k half point into the odd slot
    unsigned baseB[5];                   // per (buffer, offset between the two taps of a double-step): see kidx()
    bf16x8 ah[RB], al[RB], bh[CB], bl[CB];
    unsigned bw[8 * CB];
    f32x4 acc[RB][CB];

    __device__ __forceinline__ HaloBody16(const ConvArgs& p_, char* s) : p(p_), smem(s) {}

    // image offset (words) of step u of a stage pair: buffer (u / 9) & 1, tap u % 9
    static constexpr int tapoff(int u) { return ((u % 9) / 3) * G::RP + (u % 9) % 3; }
    static constexpr int kidx(int D) { return D == 0 ? 0 : D == 1 ? 1 : D == 2 ? 0 : D == 3 ? 0 : D == 4 ? 2 : D == 5 ? 3 : D == 6 ? 3 : D == 7 ? 4 : 3; }
    // image pieces issued per wave in double-step D: the stage's rounds in two halves, [0, NB1) at D = 0 / 5, [NB1, PPW) at D = 1 / 6
    // (3 | 2 of 5 rounds; 4 | 3 of the 7 rounds of a 512-pixel tile).  A buffer is free for two double-steps before its first reader's
    // prefetch needs it certified, so there is no third slot.
    static constexpr int NB1 = G::PPW <= 3 ? G::PPW : (G::PPW <= 5 ? 3 : 4);
    static constexpr int nB(int D) {
        return (D == 0 || D == 5) ? NB1 : (D == 1 || D == 6) ? G::PPW - NB1 : 0;
    }
    template <int K0, int K1, int BUF>
    __device__ __forceinline__ void dma_b_rounds(int st) {      // rounds [K0, K1) of stage st
        if constexpr (K0 < K1) {
            dma_b_round<K0, BUF>(st);
            dma_b_rounds<K0 + 1, K1, BUF>(st);
        }
    }

    template <int U3>
    __device__ __forceinline__ void dma_a(int Pn, const unsigned (&v)[G::APW]) {
        constexpr int slot = U3 % G::NA;
        constexpr int c = (U3 % 9) * 4 + 2 * (U3 / 9);
        const unsigned soff = (unsigned)(Pn * 36 + c) * (unsigned)p.M * 16u;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            const unsigned dst = dummyA[j] ? dstA[j] : (unsigned)(G::A_OFF + slot * G::ASZ) + dstA[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_vptr)(smem + dst), 16, v[j], soff, 0, 0);
        }
    }
    template <int K, int BUF>
    __device__ __forceinline__ void dma_b(unsigned voff, unsigned soff) {
        const unsigned dst = dummyB[K] ? dstB0[K] : (unsigned)(G::B_OFF + BUF * G::BBYTES) + dstB0[K];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_vptr)(smem + dst), 16, voff, soff, 0, 0);
    }
    template <int K, int BUF>
    __device__ __forceinline__ void dma_b_round(int st) {       // round K of stage st (>= NS: the next tile's stage 0)
        if constexpr (K < G::PPW) {
            const bool nx = st >= NS;
            dma_b<K, BUF>(nx ? vBn[K] : vB[K], nx ? 0u : (unsigned)st * 16u * hw4);
        }
    }
    template <int U3>
    __device__ __forceinline__ void dma_a_ahead() {             // A step U3 (>= 18: next pair, or the next tile's first pair)
        if constexpr (U3 < 18) dma_a<U3>(P, vA);
        else if (P < lastP) dma_a<U3 - 18>(P + 1, vA);
        else dma_a<U3 - 18>(0, vAn);
    }
    template <int D1>
    __device__ __forceinline__ void load_a(int i) {              // row block i of double-step D1 (mod 9)
        constexpr int q = (D1 % 9) % 3;
        const char* ab = smem + baseA[q];
        ah[i] = *reinterpret_cast<const bf16x8*>(ab + i * 256);
        al[i] = *reinterpret_cast<const bf16x8*>(ab + 2 * G::BM * 16 + i * 256);
    }
    template <int D1>
    __device__ __forceinline__ unsigned load_bw(int c, int j) {  // column block c, channel j of the lane's group
        constexpr int D = D1 % 9, u = 2 * D;
        return *reinterpret_cast<const unsigned*>(smem + baseB[kidx(D)] + (tapoff(u) + j * G::CHP + 16 * c) * 4);
    }
    __device__ __forceinline__ void regroup() {
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            u32x4v hv, lv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PS) {
                    hv[q] = __builtin_amdgcn_perm(bw[8 * c + 2 * q + 1], bw[8 * c + 2 * q], 0x07060302u);
                    lv[q] = __builtin_amdgcn_perm(bw[8 * c + 2 * q + 1], bw[8 * c + 2 * q], 0x05040100u);
                } else {
                    unsigned h_, l_;
                    if constexpr (H16) split2_h(__uint_as_float(bw[8 * c + 2 * q]), __uint_as_float(bw[8 * c + 2 * q + 1]), h_, l_);
                    else split2(__uint_as_float(bw[8 * c + 2 * q]), __uint_as_float(bw[8 * c + 2 * q + 1]), h_, l_);
                    hv[q] = h_;
                    lv[q] = l_;
                }
            }
            bh[c] = __builtin_bit_cast(bf16x8, hv);
            bl[c] = __builtin_bit_cast(bf16x8, lv);
        }
    }
    template <int D1>
    __device__ __forceinline__ void prefetch_all() {
#pragma unroll
        for (int i = 0; i < (HALFWIN ? RB / 2 : RB); ++i) load_a<D1>(i);       // (HALFWIN: the second half follows inside the double-step)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) bw[8 * c + j] = load_bw<D1>(c, j);
        regroup();
    }
    template <int D>
    __device__ __forceinline__ void dstep() {
        constexpr int U = 2 * D, PD = (D + 8) % 9;
#ifdef AG2V_HALO_LOOSE_WAIT      // TIMING EXPERIMENT ONLY (races): is the counted wait -- i.e. DMA latency -- what the waves park on?
        wait_vmcnt<2 * G::APW + nB(PD) + AG2V_HALO_LOOSE_WAIT>();
#else
        wait_vmcnt<2 * G::APW + nB(PD)>();
#endif
        __builtin_amdgcn_s_barrier();
        constexpr int WPG = (8 * CB) / RB;           // B words of the next double-step fetched per row-block group
#pragma unroll
        for (int g = 0; g < RB; ++g) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                // (image fragment first, weight fragment second: pixels down the accumulator registers -- see the epilogue)
                if constexpr (H16) {
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[c]), __builtin_bit_cast(f16x8, al[g]), acc[g][c], 0, 0, 0);
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl[c]), __builtin_bit_cast(f16x8, ah[g]), acc[g][c], 0, 0, 0);
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh[c]), __builtin_bit_cast(f16x8, ah[g]), acc[g][c], 0, 0, 0);
                } else {
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c], al[g], acc[g][c], 0, 0, 0);
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[c], ah[g], acc[g][c], 0, 0, 0);
                    acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[c], ah[g], acc[g][c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // The double-step's DMA pieces go out behind the first MFMA group (an LDS-DMA takes 60-180 cycles to issue).
            if (g == dma_g) {
                // image pieces: the pair's second stage (buffer 1) during double-steps 0, 1; the NEXT pair's (or tile's) first stage
                // (buffer 0, free once step 8 has been fetched: after barrier 4) during double-steps 5, 6
                if constexpr (D == 0) dma_b_rounds<0, NB1, 1>(2 * P + 1);
                if constexpr (D == 1) dma_b_rounds<NB1, G::PPW, 1>(2 * P + 1);
                if constexpr (D == 5) dma_b_rounds<0, NB1, 0>(2 * P + 2);
                if constexpr (D == 6) dma_b_rounds<NB1, G::PPW, 0>(2 * P + 2);
                if (HALFWIN && lateA) dma_a_ahead<U + 4>();
                else dma_a_ahead<U + 6>();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == dma_g + 1) {
                if (HALFWIN && lateA) dma_a_ahead<U + 5>();
                else dma_a_ahead<U + 7>();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (HALFWIN) {         // row block g + RB/2 of this double-step, then row blocks 0 .. RB/2 - 1 of the next
                if (g < RB / 2) load_a<D>(g + RB / 2);
                else load_a<D + 1>(g - RB / 2);
            } else {
                load_a<D + 1>(g);
            }
#pragma unroll
            for (int j = 0; j < WPG; ++j) {
                const int w = g * WPG + j;
                bw[w] = load_bw<D + 1>(w >> 3, w & 7);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        regroup();
        __builtin_amdgcn_sched_barrier(0);
    }

    struct Tile { int n_img, y0, x0, m0; };
    __device__ __forceinline__ Tile describe(int v, unsigned (&va)[G::APW], unsigned (&vb)[G::PPW]) {
        Tile t = {0, 0, 0, 0};
        const bool live = v >= 0;
        const int lid = live ? xcd_remap(v, p.tiles_m * p.tiles_n) : 0;
        const int tile_m = lid % p.tiles_m, tn = lid / p.tiles_m;
        const int tcols = p.OW / TW, tpi = (p.OH / TH) * tcols;
        t.n_img = tn / tpi;
        const int trem = tn - t.n_img * tpi;
        t.y0 = (trem / tcols) * TH;
        t.x0 = (trem % tcols) * TW;
        t.m0 = tile_m * G::BM;
        const unsigned lo_plane = (unsigned)(p.KG >> 3) * (unsigned)p.M * 16u;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            const int e = (wave + 8 * j) * 64 + lane;
            const int plane = e / (2 * G::BM), kg = (e / G::BM) & 1, m = e % G::BM;
            va[j] = (live && !dummyA[j] && t.m0 + m < p.M) ? (unsigned)plane * lo_plane + (unsigned)(kg * p.M + t.m0 + m) * 16u : kOut;
        }
#pragma unroll
        for (int k = 0; k < G::PPW; ++k) {
            const int ql = (wave + 8 * k) * 64 + lane;                       // quad position in the LDS image (with the skew hole)
            const bool hole = ql >= 2 * G::CHP && ql < 2 * G::CHP + G::SKEW / 4;
            const int q = ql < 2 * G::CHP ? ql : ql - G::SKEW / 4;
            const int ch = q / (G::CHP / 4), w = q % (G::CHP / 4);
            const int row = w / (G::RP / 4), cq = w % (G::RP / 4);
            const int y = t.y0 - 1 + row, x = t.x0 - 4 + 4 * cq;
            const bool ok = live && !dummyB[k] && !hole && ql < G::BQ && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            vb[k] = ok ? (unsigned)(((t.n_img * p.C + ch) * p.H + y) * p.W + x) * 4u : kOut;
        }
        return t;
    }

    __device__ __forceinline__ void run() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        ars = make_rsrc(p.a_src, p.a_bytes);
        brs = make_rsrc(p.b_src, p.b_bytes);
        hw4 = (unsigned)(p.H * p.W) * 4u;
        NS = p.C / 16;
        lastP = NS / 2 - 1;
#pragma unroll
        for (int j = 0; j < G::APW; ++j) {
            dummyA[j] = wave + 8 * j >= G::APS;
            dstA[j] = dummyA[j] ? (unsigned)G::DUMMY_OFF : (unsigned)(wave + 8 * j) * 1024u;
        }
#pragma unroll
        for (int k = 0; k < G::PPW; ++k) {
            dummyB[k] = wave + 8 * k >= G::NPB;
            dstB0[k] = dummyB[k] ? (unsigned)G::DUMMY_OFF : (unsigned)(wave + 8 * k) * 1024u;
        }
        {   // the bias vector -> LDS, zero-padded to whole row tiles (epilogue; the first tile's barrier publishes it)
            float* lb = reinterpret_cast<float*>(smem + G::BIAS_OFF);
            float* ls = reinterpret_cast<float*>(smem + G::SUM_OFF);
            for (int i = tid; i < p.tiles_m * G::BM; i += 512) {
                lb[i] = (p.bias && i < p.M) ? p.bias[i] : 0.f;
                ls[i] = 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const int wm = wave / WN, wn = wave % WN;
        const int wm0 = wm * EM * 32;
        const int pb = wn * 16 * CB, ty = pb / TW, tx = pb % TW;
        const int l15 = lane & 15, kq = lane >> 4, kg = kq & 1, sub = kq >> 1;
        // piece a = wave (+ 8): 64 consecutive rows of one (plane, k group); rows [64 h, 64 h + 64) are the early (h even) / late (h odd)
        // row blocks of the waves that own rows [128 (h / 2), + 128)
        lateA = HALFWIN && (((wave * 64) % G::BM) / 64) % 2 == 1;
        dma_g = (wave >= 4 && getenv_stagger()) ? RB / 2 : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) baseA[q] = (unsigned)(G::A_OFF + (2 * q + sub) * G::ASZ + kg * G::BM * 16 + (wm0 + l15) * 16);
        const int lb = (kg * G::G8 + ty * G::RP + tx + l15 + 3) * 4;
        const int d_next = 4, d_row = (G::RP - 2) * 4, d_stage = G::BBYTES - (2 * G::RP + 2) * 4;
        baseB[0] = (unsigned)(G::B_OFF + lb + sub * d_next);
        baseB[1] = (unsigned)(G::B_OFF + lb + sub * d_row);
        baseB[2] = (unsigned)(G::B_OFF + lb + sub * d_stage);
        baseB[3] = (unsigned)(G::B_OFF + G::BBYTES + lb + sub * d_next);
        baseB[4] = (unsigned)(G::B_OFF + G::BBYTES + lb + sub * d_row);
        asm volatile("" : "+v"(baseA[0]), "+v"(baseA[1]), "+v"(baseA[2]), "+v"(baseB[0]), "+v"(baseB[1]), "+v"(baseB[2]), "+v"(baseB[3]), "+v"(baseB[4]));
        const int ntiles = p.tiles_m * p.tiles_n, stride = gridDim.x;
        int v = blockIdx.x;
        Tile nxt = describe(v, vAn, vBn);
        P = lastP;
        {   // first tile: its stage 0 (as "the next tile's") and A steps 0..5
            const int keepNS = NS;
            NS = 0;                       // every stage index counts as "beyond": dma_b_round takes vBn with offset 0
            dma_b_rounds<0, G::PPW, 0>(0);
            NS = keepNS;
        }
        dma_a<0>(0, vAn); dma_a<1>(0, vAn); dma_a<2>(0, vAn); dma_a<3>(0, vAn);
        if (!(HALFWIN && lateA)) { dma_a<4>(0, vAn); dma_a<5>(0, vAn); }       // (late-row pieces run two double-steps ahead, not three)
        const int plane = p.OH * p.OW;
        const float* aux0 = p.residual ? p.residual : p.mask;
        const float ninf = -__builtin_huge_valf();
        const float rmul = p.residual ? 1.f : 0.f;
        const float relu_floor = p.relu ? 0.f : ninf;
        const float mask_thr = (!p.residual && p.mask) ? 0.f : ninf;
        const bool osplit = p.out_split != 0;
        bool first = true, full_prev = false;
        while (true) {
            const Tile cur = nxt;
#pragma unroll
            for (int j = 0; j < G::APW; ++j) vA[j] = vAn[j];
#pragma unroll
            for (int k = 0; k < G::PPW; ++k) vB[k] = vBn[k];
            v += stride;
            const bool more = v < ntiles;
            nxt = describe(more ? v : -1, vAn, vBn);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][c][r] = 0.f;
            // this tile's stage 0 and A steps 0..5 (issued by the previous tile's tail, or just above) have landed.  They are OLDER than
            // the previous tile's RB * CB epilogue stores, which may stay in flight (vmcnt counts in issue order); a tile whose last row block
            // is partly outside M issues fewer stores, so it waits for everything.
            if (first || !full_prev) wait_vmcnt<0>();
            else wait_vmcnt<RB * CB>();
            first = false;
            full_prev = cur.m0 + G::BM <= p.M;
            // the tile's group scale (epilogue), through the scalar cache: a vector load would sit in the vmcnt queue
            float gsv = 1.f;
            if (p.gscale) {
                const float* gp = p.gscale + cur.n_img / p.group_n;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gsv) : "s"(gp) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            prefetch_all<0>();
            for (P = 0; P <= lastP; ++P) {
                dstep<0>(); dstep<1>(); dstep<2>(); dstep<3>(); dstep<4>(); dstep<5>(); dstep<6>(); dstep<7>(); dstep<8>();
            }
            // ---- epilogue.  The MFMAs take the IMAGE fragment as their first operand and the weight fragment as their second (the two
            // fragment layouts are the same: index lane & 15, k group lane >> 4), so tile (i, c) register r holds channel 16 i + (lane & 15),
            // pixel 16 c + 4 (lane >> 4) + r: every lane owns FOUR CONSECUTIVE PIXELS of one output channel as it stands -- a 16-byte store
            // per tile and lane (16 channel runs of 64 bytes per instruction), the bias is a per-lane value, residual / mask words are
            // 16-byte loads at the same address pattern.  (Rounds 4-5 multiplied the other way round -- channels down the registers -- and
            // paid a v_permlane16_swap + two DPP transpose stages + two bias bpermutes per register pair for the same stores: ~70 VALU
            // instructions per accumulator tile pair instead of ~20, with both waves of a SIMD in their epilogues at once and the matrix
            // pipe idle: 11 % of a 128-channel tile's time)
            const int pq = 4 * (lane >> 4);
            const size_t pxo = (size_t)(cur.y0 + ty) * p.OW + (cur.x0 + tx + pq);
            const size_t pix = (size_t)cur.n_img * p.M * plane + pxo;
            float* obase = p.out + (size_t)cur.n_img * p.out_nstride + pxo;
            // (multi-output launch: this row tile's own tensor; rows are addressed by their absolute index below)
            if (p.out_mstride) obase += (long long)(cur.m0 / G::BM) * p.out_mstride - (long long)cur.m0 * plane;
            const float* aux = aux0 ? aux0 + pix : nullptr;
            const float gs = gsv * (p.oscale != 0.f ? p.oscale : 1.f);
            // The bias vector sits in LDS since the start of the launch (zeros without one, and beyond row M) and the group scale came
            // through the scalar cache at the top of the tile: an epilogue without a residual / mask issues no global load at all, so
            // nothing in it waits on the vmcnt queue, where the next tile's first DMA pieces are in flight.
            const float* lbias = reinterpret_cast<const float*>(smem + G::BIAS_OFF) + (cur.m0 + wm0 + l15);
            float* lsums = reinterpret_cast<float*>(smem + G::SUM_OFF) + (cur.m0 + wm0 + l15);
            const bool want_sums = p.chan_sums != nullptr;
            // AUX: a residual / mask tensor is read; SPLIT: the output is written as split words -- both uniform over the launch: one of
            // four straight-line instances runs (a select per VALUE on these flags was a scalar branch per value: 64 per lane and tile)
            auto epilogue = [&](auto AUX, auto SPLIT) {
                constexpr bool kAux = decltype(AUX)::value, kSplit = decltype(SPLIT)::value;
                auto finish = [&](float a, float b_, float av) -> float {
                    float x = fmaf(a, gs, b_);
                    if constexpr (kAux) x = fmaf(av, rmul, x);
                    x = fmaxf(x, relu_floor);
                    if constexpr (kAux) x = (av > mask_thr) ? x : 0.f;
                    return x;
                };
                auto word = [&](float x) -> float {          // the stored form of a value
                    if constexpr (kSplit) return pack_split(x);
                    else return x;
                };
                // ConvArgs::chan_sums: the row's sum over this wave's pixels (the four lanes that share lane & 15), added to the
                // workgroup's LDS record of the launch's rows; the records go to memory once, when the workgroup has run out of tiles
                auto row_sum = [&](int i, const float4 (&x4)[CB]) {
                    float sx = 0.f;
#pragma unroll
                    for (int c = 0; c < CB; ++c) sx += (x4[c].x + x4[c].y) + (x4[c].z + x4[c].w);
                    sx += __shfl_xor(sx, 16);
                    sx += __shfl_xor(sx, 32);
                    if (lane < 16) atomicAdd(lsums + 16 * i, sx);
                };
                if constexpr (!kAux) {
                    // (most forward launches) row block by row block: results, then stores.  Until round 6 this case issued placeholder
                    // loads of one cached line to keep a single branch-free instruction stream: two exposed round trips per tile.
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        const int m = cur.m0 + wm0 + i * 16 + l15;
                        const float b_ = lbias[16 * i];
                        float4 o[CB];
#pragma unroll
                        for (int c = 0; c < CB; ++c)
                            o[c] = make_float4(finish(acc[i][c][0], b_, 0.f), finish(acc[i][c][1], b_, 0.f), finish(acc[i][c][2], b_, 0.f),
                                               finish(acc[i][c][3], b_, 0.f));
                        if (want_sums) row_sum(i, o);
#pragma unroll
                        for (int c = 0; c < CB; ++c)
                            if (m < p.M)
                                *reinterpret_cast<float4*>(obase + (size_t)m * plane + 16 * c) = make_float4(word(o[c].x), word(o[c].y), word(o[c].z), word(o[c].w));
                    }
                } else {
                    // Residual / mask words are fetched by loads whose rows beyond M are clamped (no branch around a load), in two batches
                    // of RB / 2 row blocks, ordered
                    //   loads 1 | results 1 | loads 2 | stores 1 | results 2 | stores 2
                    // so that no load is ever younger than a store it does not depend on: vmcnt counts in issue order, and a load behind
                    // stores makes the wait for it cover those stores as well.  Two batches instead of one keep the epilogue within the
                    // register budget of the main loop.
                    constexpr int HB = RB / 2;
                    float4 axs[HB][CB], ov[HB][CB];
                    auto loads = [&](int h) {
#pragma unroll
                        for (int k = 0; k < HB; ++k) {
                            const int mc = min(cur.m0 + wm0 + (h * HB + k) * 16 + l15, p.M - 1);
#pragma unroll
                            for (int c = 0; c < CB; ++c) axs[k][c] = *reinterpret_cast<const float4*>(aux + (size_t)mc * plane + 16 * c);
                        }
                    };
                    auto results = [&](int h) {
#pragma unroll
                        for (int k = 0; k < HB; ++k) {
                            const int i = h * HB + k;
                            const float b_ = lbias[16 * i];
#pragma unroll
                            for (int c = 0; c < CB; ++c) {
                                const float4 a4 = axs[k][c];
                                ov[k][c] = make_float4(finish(acc[i][c][0], b_, a4.x), finish(acc[i][c][1], b_, a4.y), finish(acc[i][c][2], b_, a4.z),
                                                       finish(acc[i][c][3], b_, a4.w));
                            }
                            if (want_sums) row_sum(i, ov[k]);
#pragma unroll
                            for (int c = 0; c < CB; ++c) ov[k][c] = make_float4(word(ov[k][c].x), word(ov[k][c].y), word(ov[k][c].z), word(ov[k][c].w));
                        }
                    };
                    auto stores = [&](int h) {
#pragma unroll
                        for (int k = 0; k < HB; ++k) {
                            const int m = cur.m0 + wm0 + (h * HB + k) * 16 + l15;
#pragma unroll
                            for (int c = 0; c < CB; ++c)
                                if (m < p.M) *reinterpret_cast<float4*>(obase + (size_t)m * plane + 16 * c) = ov[k][c];
                        }
                    };
                    loads(0);
                    __builtin_amdgcn_sched_barrier(0);
                    results(0);
                    __builtin_amdgcn_sched_barrier(0);
                    loads(1);
                    __builtin_amdgcn_sched_barrier(0);
                    stores(0);
                    __builtin_amdgcn_sched_barrier(0);
                    results(1);
                    __builtin_amdgcn_sched_barrier(0);
                    stores(1);
                }
            };
            typedef std::integral_constant<bool, true> Yes;
            typedef std::integral_constant<bool, false> No;
            if (aux0 == nullptr) {
                if (osplit) epilogue(No{}, Yes{}); else epilogue(No{}, No{});
            } else {
                if (osplit) epilogue(Yes{}, Yes{}); else epilogue(Yes{}, No{});
            }
            if (!more) break;
        }
        wait_vmcnt<0>();
        if (p.chan_sums) {      // the workgroup's row sums -> memory (one float atomic per row and workgroup)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float* ls = reinterpret_cast<const float*>(smem + G::SUM_OFF);
            for (int i = tid; i < p.M; i += 512) {
                const float v_ = ls[i];
                if (v_ != 0.f) atomicAdd(p.chan_sums + i, v_);
            }
        }
    }
};
#endif

// PS: the gathered tensor holds split words (false: fp32 -- the wave that multiplies splits its own fragment)
template <int WM, int WN, int EM, int TW, int TH, bool PS>
__global__ __launch_bounds__(512) void conv_halo_bf16x3_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) char smem[HaloCfg<WM, WN, EM, TW, TH>::LDS_BYTES];
    HaloBody<WM, WN, EM, TW, TH, PS> body(p, smem);
    body.run();
#endif
}
// the 16x16x32 form (HaloBody16); CB: 16-pixel column blocks per wave (HaloCfg)
template <int WM, int WN, int EM, int TW, int TH, bool PS, int CB = 2>
__global__ __launch_bounds__(512) void conv_halo16_bf16x3_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) char smem[HaloCfg<WM, WN, EM, TW, TH, true, CB>::LDS_BYTES];
    HaloBody16<WM, WN, EM, TW, TH, PS, false, CB> body(p, smem);
    body.run();
#endif
}

// ... on the fp16 3-term split (fp32 gathered tensor, ConvArgs::h16)
template <int WM, int WN, int EM, int TW, int TH, int CB = 2>
__global__ __launch_bounds__(512) void conv_halo16_f16x3_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) char smem[HaloCfg<WM, WN, EM, TW, TH, true, CB>::LDS_BYTES];
    HaloBody16<WM, WN, EM, TW, TH, false, true, CB> body(p, smem);
    body.run();
#endif
}

// ---- host side
// Geometry of the pixel patch for a BN-pixel tile on an OH x OW map: the widest TW in {128, 64, 32} that divides OW with TH = BN / TW
// dividing OH and leaving at least one row outside the patch (a patch as tall as the image would make pieces that are all padding).
static bool halo_geometry(int BN, int OH, int OW, int& TW, int& TH) {
    for (int tw = 128; tw >= 32; tw >>= 1) {
        if (tw > BN || OW % tw) continue;
        const int th = BN / tw;
        if (OH % th || th >= OH) continue;
        TW = tw; TH = th;
        return true;
    }
    return false;
}


// ---- diagnostic / A-B switches of the launchers, read from the environment ONCE (not per launch: getenv beside a concurrent putenv is
// not thread-safe, and the autograd thread launches kernels while the main thread may be editing os.environ -- ADVICE r04) and again
// only when ag2v_halo_reload_env() is called (tests and tools/halo_bench.py call it after changing the variables, with no launch in
// flight).  None of them changes results beyond the summation order of the kernel family selected.
struct HaloSwitches { int no_halo, mfma32, min_tiles, no_wgrad_halo, no_wide; };
static HaloSwitches read_halo_switches() {
    HaloSwitches h;
    h.no_halo = getenv("AG2V_NO_HALO") != nullptr;
    h.mfma32 = getenv("AG2V_HALO_MFMA32") != nullptr;
    h.no_wide = getenv("AG2V_NO_HALO_WIDE") != nullptr;            // (tests: the 64-row layers on the 256-pixel tile, as before round 6)
    h.no_wgrad_halo = getenv("AG2V_NO_WGRAD_HALO") != nullptr;      // (conv_wgrad_halo.hip: weight gradients on the previous kernels)
    const char* mt = getenv("AG2V_HALO_MIN_TILES");                 // (tests run small shapes through these kernels)
    h.min_tiles = mt ? atoi(mt) : 384;                              // small maps: the split-K kernels fill the chip better
    return h;
}
static HaloSwitches g_halo_sw = read_halo_switches();
extern "C" int ag2v_halo_reload_env(void) {
    g_halo_sw = read_halo_switches();
    return AG2V_OK;
}
extern "C" int ag2v_halo_min_tiles(void) { return g_halo_sw.min_tiles; }
bool wgrad_halo_switched_off() { return g_halo_sw.no_wgrad_halo != 0; }

// 64-row layers (up_3, the flow network's full-resolution layers): 512 pixels per workgroup -- 64 per wave (HaloCfg CB = 4) -- as 128 x 4 or
// 64 x 8 patches, when the map allows it and the 16x16x32 form is on
static bool halo_geometry_wide(int OH, int OW, int& TW, int& TH) {
    if (g_halo_sw.mfma32 || g_halo_sw.no_wide) return false;
    for (int tw = 128; tw >= 64; tw >>= 1) {
        if (OW % tw) continue;
        const int th = 512 / tw;
        if (OH % th || th >= OH) continue;
        TW = tw; TH = th;
        return true;
    }
    return false;
}

// (A 256-row tile -- 128 x 1 / 64 x 2 / 32 x 4 pixels, 3x row halo -- was built and measured in round 4: the 128-row tile with 256 pixels as
// 2 x 128 (2x row halo, 8 KB of weights per k-step) beat it on every shape, +1.9 % at C = 384, +4.5 % at C = 128; removed in round 6.)

// Can this launch (as the callers of launch_igemm_bf16x3 describe it) run on the halo kernels?
bool igemm_halo_ok(const ConvArgs& a) {
    if (g_halo_sw.no_halo || !a.tap_major || a.f32 || a.b_bytes == 0) return false;
    // fp16 split: fp32 gathered tensor, plain output, the 16x16x32 form only
    // (not in deterministic mode: there the flow network keeps ONE kernel family for every batch size, so that the frame-batched and
    // the sequential passes agree to the last bit in their flows -- tests/test_fullsize_gpu.py::test_full_size_deterministic_mode)
    if (a.h16 && (a.b_split || a.out_split || a.det || g_halo_sw.mfma32)) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.pad_w != 1 || a.OH != a.H || a.OW != a.W) return false;
    if (a.C % 32 || a.W % 4 || (a.o_mul != 1 && a.o_mul != 0) || a.o_add_y || a.o_add_x) return false;
    if ((a.o_H && a.o_H != a.OH) || (a.o_W && a.o_W != a.OW)) return false;
    if (a.residual == a.out && a.residual) return false;          // accumulating launches stay on the split-K capable kernels
    if (a.M > 2048) return false;                                 // the bias vector's place in LDS (HaloCfg::BIAS_ROWS)
    const int bm = a.M <= 64 ? 64 : 128;
    if (a.out_mstride && (bm != 128 || a.M % 128 || a.residual || a.mask)) return false;      // per-row-tile outputs: whole 128-row tiles
    int tw, th;
    const bool wide = bm == 64 && halo_geometry_wide(a.OH, a.OW, tw, th);
    const int bn = wide ? 512 : 256;
    if (!wide && !halo_geometry(bn, a.OH, a.OW, tw, th)) return false;
    // (the threshold counts 256-pixel tiles for the wide form too: WHICH launches take these kernels does not depend on the tile width --
    // the frame-batched and the sequential generator keep choosing the same family for a layer)
    (void)bn;
    const long long tiles = (long long)ceil_div(a.M, bm) * (a.NG / 256);
    return tiles >= g_halo_sw.min_tiles;
}

bool halo_chan_sums_ok(ConvArgs a) {
    if (a.o_mul == 0) { a.o_mul = 1; a.o_add_y = 0; a.o_add_x = 0; a.o_H = a.OH; a.o_W = a.OW; }      // (as launch_igemm_bf16x3 completes them)
    if (a.out_nstride == 0) a.out_nstride = (long long)a.M * a.o_H * a.o_W;
    return !g_halo_sw.mfma32 && igemm_halo_ok(a);
}

int launch_igemm_halo(ConvArgs& a, hipStream_t st) {
    if (a.out_nstride == 0) a.out_nstride = (long long)a.M * a.OH * a.OW;       // (direct callers: the dense output)
    const int bm = a.M <= 64 ? 64 : 128;
    int tw, th;
    const bool wide = bm == 64 && halo_geometry_wide(a.OH, a.OW, tw, th);
    const int bn = wide ? 512 : 256;
    if (!wide && !halo_geometry(bn, a.OH, a.OW, tw, th)) return AG2V_EINVAL;
    a.tiles_m = ceil_div(a.M, bm);
    a.tiles_n = a.NG / bn;
    a.splitk = 1;
    // persistent workgroups, one per CU (the LDS footprint admits no second one), a multiple of 8 so that a workgroup's tiles stay on
    // one XCD's share of the logical tile order (xcd_remap)
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v & ~7;
        if (n_cu < 8) n_cu = 8;
    }
    const long long ntiles = (long long)a.tiles_m * a.tiles_n;
    const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu));
    const bool mf16 = !g_halo_sw.mfma32;          // (A/B aid: the 32x32x16 form)
    if (!(mf16 || a.h16 || wide)) a.chan_sums = nullptr;      // (the 32x32x16 form has no channel sums: the caller sums)
    note_chan_sums_done(a.chan_sums != nullptr);
#define AG2V_HALO(WM_, WN_, EM_, TW_, TH_)                                                                                       \
    do {                                                                                                                         \
        if (a.h16) {                                                                                                             \
            set_last_conv_kernel("conv_halo16_f16x3_kernel<" #WM_ ", " #WN_ ", " #EM_ ", " #TW_ ", " #TH_ ">");                    \
            hipLaunchKernelGGL((conv_halo16_f16x3_kernel<WM_, WN_, EM_, TW_, TH_>), grid, dim3(512), 0, st, a);                    \
        } else if (mf16 && a.b_split) {                                                                                                 \
            set_last_conv_kernel("conv_halo16_bf16x3_kernel<" #WM_ ", " #WN_ ", " #EM_ ", " #TW_ ", " #TH_ ", true>");            \
            hipLaunchKernelGGL((conv_halo16_bf16x3_kernel<WM_, WN_, EM_, TW_, TH_, true>), grid, dim3(512), 0, st, a);             \
        } else if (mf16) {                                                                                                       \
            set_last_conv_kernel("conv_halo16_bf16x3_kernel<" #WM_ ", " #WN_ ", " #EM_ ", " #TW_ ", " #TH_ ", false>");           \
            hipLaunchKernelGGL((conv_halo16_bf16x3_kernel<WM_, WN_, EM_, TW_, TH_, false>), grid, dim3(512), 0, st, a);            \
        } else if (a.b_split) {                                                                                                         \
            set_last_conv_kernel("conv_halo_bf16x3_kernel<" #WM_ ", " #WN_ ", " #EM_ ", " #TW_ ", " #TH_ ", true>");              \
            hipLaunchKernelGGL((conv_halo_bf16x3_kernel<WM_, WN_, EM_, TW_, TH_, true>), grid, dim3(512), 0, st, a);               \
        } else {                                                                                                                 \
            set_last_conv_kernel("conv_halo_bf16x3_kernel<" #WM_ ", " #WN_ ", " #EM_ ", " #TW_ ", " #TH_ ", false>");             \
            hipLaunchKernelGGL((conv_halo_bf16x3_kernel<WM_, WN_, EM_, TW_, TH_, false>), grid, dim3(512), 0, st, a);              \
        }                                                                                                                        \
    } while (0)
#define AG2V_HALO_WIDE(TW_, TH_)                                                                                                 \
    do {                                                                                                                         \
        if (a.h16) {                                                                                                             \
            set_last_conv_kernel("conv_halo16_f16x3_kernel<1, 8, 2, " #TW_ ", " #TH_ ", 4>");                                     \
            hipLaunchKernelGGL((conv_halo16_f16x3_kernel<1, 8, 2, TW_, TH_, 4>), grid, dim3(512), 0, st, a);                      \
        } else if (a.b_split) {                                                                                                  \
            set_last_conv_kernel("conv_halo16_bf16x3_kernel<1, 8, 2, " #TW_ ", " #TH_ ", true, 4>");                              \
            hipLaunchKernelGGL((conv_halo16_bf16x3_kernel<1, 8, 2, TW_, TH_, true, 4>), grid, dim3(512), 0, st, a);               \
        } else {                                                                                                                 \
            set_last_conv_kernel("conv_halo16_bf16x3_kernel<1, 8, 2, " #TW_ ", " #TH_ ", false, 4>");                             \
            hipLaunchKernelGGL((conv_halo16_bf16x3_kernel<1, 8, 2, TW_, TH_, false, 4>), grid, dim3(512), 0, st, a);              \
        }                                                                                                                        \
    } while (0)
    if (wide) {
        if (tw == 128) AG2V_HALO_WIDE(128, 4);
        else AG2V_HALO_WIDE(64, 8);
    } else if (bm == 128) {
        if (tw == 128) AG2V_HALO(1, 8, 4, 128, 2);
        else if (tw == 64) AG2V_HALO(1, 8, 4, 64, 4);
        else AG2V_HALO(1, 8, 4, 32, 8);
    } else {
        if (tw == 128) AG2V_HALO(1, 8, 2, 128, 2);
        else if (tw == 64) AG2V_HALO(1, 8, 2, 64, 4);
        else AG2V_HALO(1, 8, 2, 32, 8);
    }
#undef AG2V_HALO
#undef AG2V_HALO_WIDE
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
