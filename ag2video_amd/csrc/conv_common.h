// Shared declarations of the convolution kernels (conv_igemm.hip: exact fp32 MFMA; conv_bf16x3.hip: split bf16 MFMA).
#pragma once
#include "common.h"

struct ConvArgs {
    const float* a_src;     // igemm: At [KG][M] (k-major weights; tap_major: bf16 hi / lo planes [KG/8][M][8]) | wgrad: dy
    const float* b_src;     // gathered tensor [N, C, H, W]
    float* out;
    const float* bias;      // [M] or null (igemm)
    const float* residual;  // same shape as out, or null (igemm)
    int N, C, H, W;         // gathered tensor
    int K, R, S, OH, OW;    // output channels / filter / GEMM pixel space (igemm);  filters / dy image (wgrad)
    int stride, pad;        // pad = padding along y
    int pad_w;              // padding along x (differs from pad only for the parity sub-filters of a strided dgrad)
    int o_mul, o_add_y, o_add_x, o_H, o_W;  // output placement: (oy,ox) -> (oy*o_mul+o_add_y, ox*o_mul+o_add_x) in [o_H,o_W]
    int M, NG, KG;          // GEMM dims
    int tiles_m, tiles_n;
    unsigned a_bytes, b_bytes;  // byte sizes of a_src / b_src (bf16x3 kernels address them through 32-bit buffer offsets)
    int splitk, kchunk;     // igemm: K-range splits (grid.y) in K-tiles; wgrad: pixel-range splits, pixels per split
    const float* mask;      // igemm epilogue: out = (mask > 0) ? out : 0 (same shape as out) -- the ReLU backward of the conv's input
    int relu;               // igemm epilogue: out = max(out, 0)
    int tap_major;          // igemm bf16x3: a_src rows are ordered by tap_major_row() instead of kg = c * R*S + r*S + s and stored
                            // as two bf16 planes, eight consecutive rows per output row ([KG/8][M][8]); set by igemm_tap_major() for
                            // C % 32 == 0 (every 32-wide K-tile then has one filter tap)
    int diag_noprio;        // tuning aid: the MFMA waves of the wgrad v kernels keep the default issue priority
    int accumulate;         // wgrad: add onto the existing dW (no zeroing; atomics) -- the weight is used by several forward calls
    const float* gscale;    // igemm epilogue (nullable): the accumulated product of sample n is multiplied by gscale[n / group_n]
    int group_n;            //   before bias / residual -- spectral norm's 1/sigma_t of frame group t (frame-batched generator)
    int groups, gq, spg;    // wgrad: `groups` consecutive sample groups of gq pixels each, spg pixel-range splits per group (grid =
    long long out_gstride;  //   tiles x groups x spg); group g writes dW at out + g * out_gstride.  groups == 1: the plain form.
    int out_split;          // igemm bf16x3 epilogue: store the finished values as split words (forward with relu: the activation is
                            //   read by bf16x3 convolutions only)
    int f32;                // wave-specialised implicit-GEMM kernels on exact fp32 operands (conv_igemm_f32_ws_kernel)
    int a_split;            // weight-gradient bf16x3 kernels: dy (a_src) holds split words
    int b_split;            // bf16x3 kernels: b_src holds split words ((bf16 hi << 16) | bf16 lo per value, pack_split()) instead of fp32
    int h16;                // igemm: FP16 3-term split (ag2v_conv_opts::precision == 2): x = hi + lo with fp16 hi / lo (11 + 11 bits) on
                            //   v_mfma_f32_32x32x16_f16 -- ~5e-6 relative (fp32-class; bf16: ~1.5e-5) for operands of O(1) magnitude;
                            //   the packed weights are scaled by kH16WScale (a power of two), the epilogue multiplies by oscale
    float oscale;           // igemm epilogue: factor on the accumulated product (0 = 1)
    int det;                // deterministic reduction (ag2v_conv_opts::deterministic): igemm: no split-K (launchers); wgrad: pixel-range
                            //   split number `split` stores its partial tile to out + split * M * NG (a caller workspace slab, plain
                            //   stores) and launch_wgrad_det_reduce() adds the slabs in split order -- no floating-point atomics
    long long out_nstride;  // igemm epilogue: floats between consecutive samples of `out` (0 = M * o_H * o_W, the dense tensor; set
                            //   by launch_igemm).  Larger: out is a channel slice of a wider [N, Ktot, H, W] buffer
                            //   (ag2v_conv_opts::out_nstride: data gradients written straight into the concatenation their consumer
                            //   reads); residual / mask keep the dense layout.  Such launches never split K.
    long long out_mstride;  // halo kernels' epilogue (0 = off): floats between the outputs of consecutive 128-ROW TILES -- row tile i of the
                            //   launch is its own dense [N, 128, H, W] tensor at out + i * out_mstride (out_nstride = 128 * H * W): ONE
                            //   launch computes several convolutions of the same input (the SPADE mlp_shared group of a block,
                            //   ag2v_conv_opts::out_mstride) and every consumer still reads a contiguous tensor
    float* chan_sums;       // halo16 kernels (0 = off): += per-row sums of the values the epilogue writes (ag2v_conv_opts::out_chan_sums)
#ifdef AG2V_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/ubench/igemm_stamps.hip): per-wave phase clocks
#endif
};

// Division by the filter footprint: the shapes on the path get constant divisors.
__device__ __forceinline__ int div_small(int v, int d) {
    if (d == 9) return v / 9;
    if (d == 16) return v >> 4;
    if (d == 1) return v;
    if (d == 3) return v / 3;
    if (d == 4) return v >> 2;
    if (d == 2) return v >> 1;
    return v / d;
}
__device__ __forceinline__ void decode_crs(int kg, int RS, int S, int& c, int& r, int& s) {
    c = div_small(kg, RS);
    const int rs = kg - c * RS;
    r = div_small(rs, S);
    s = rs - r * S;
}


// ---------------------------------------------------------------- bf16x3 split-precision helpers
// x = hi + lo + O(2^-18 |x|) with hi = bf16(x), lo = bf16(x - hi);  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (the dropped
// a_lo*b_lo term is <= 2^-18 |a b|), i.e. three bf16 MFMAs (exact 16-bit products, fp32 accumulation) reproduce an fp32
// product to ~1e-5 relative -- two orders inside the 1e-3 parity bound -- at 16/3 of the fp32 MFMA rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    float d0, d1;
#ifndef AG2V_SPLIT_DOT2
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    // two v_sub_f32, not one v_pk_add_f32: packed f32 VALU beside MFMAs costs ~13 extra cycles per instruction
    // (MI355X_MICROARCH.md constants table) and the compiler SLP-packs a plain v - hf
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x0), "v"(hf[0]));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x1), "v"(hf[1]));
#else
    // Measured and NOT used: x - hi straight from the packed pair with v_dot2c_f32_bf16 (x += h.lo * -1 + h.hi * 0; 4 VALU per
    // pair instead of 6).  The c3 step went from 885 to 959 ms: the dot instruction does not issue beside the MFMAs the way
    // plain VALU does.
    const bf16x2 m0 = {(__bf16)-1.0f, (__bf16)0.0f};
    const bf16x2 m1 = {(__bf16)0.0f, (__bf16)-1.0f};
    d0 = __builtin_amdgcn_fdot2_f32_bf16(h, m0, x0, false);
    d1 = __builtin_amdgcn_fdot2_f32_bf16(h, m1, x1, false);
#endif
    const f32x2 dv = {d0, d1};
    const bf16x2 l = __builtin_convertvector(dv, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// ---- fp16 3-term split (forward passes on O(1)-scaled operands: the flow network, whose output feeds the warp).  fp16 carries
// 11 significand bits against bf16's 8: hi + lo covers 22 bits and hi*hi + hi*lo + lo*hi is accurate to ~2^-21 -- measured 4.5e-6 of
// rms(out) on K = 4608 dot products, BELOW a sequential fp32 fmaf chain's 6.5e-6, where the bf16 split has 1.6e-5
// (tools/ubench/f16x3_probe.hip).  The price is range: below ~1e-2 the lo term is an fp16 subnormal and the accuracy decays (8e-5 at
// 1e-3), so the weights are packed times kH16WScale = 32 (kaiming-scale filters become O(1); exact, undone by ConvArgs::oscale) and
// the mode is only used where activations are normalised (ag2v_conv_opts::precision == 2).  Same MFMA rate as bf16.
#define AG2V_H16_WSCALE 32.0f
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_h(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    float d0, d1;
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x0), "v"(hf[0]));      // (two v_sub_f32: see split2)
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x1), "v"(hf[1]));
    const f32x2 dv = {d0, d1};
    const f16x2 l = __builtin_convertvector(dv, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// fp32 -> packed (fp16 hi << 16) | fp16 lo of x * AG2V_H16_WSCALE: the pre-split weight operand of the h16 kernels
__device__ __forceinline__ float pack_split_h(float x) {
    x *= AG2V_H16_WSCALE;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const unsigned w = ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
    return __uint_as_float(w);
}

// one fp32 -> packed word (hi bf16 in the upper half, lo bf16 in the lower half); used for the pre-split weight operand
__device__ __forceinline__ float pack_split(float x) {
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    const unsigned w = ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
    return __uint_as_float(w);
}


// Raw buffer loads: a wave-uniform descriptor in SGPRs + a 32-bit byte offset per lane (no 64-bit address VGPRs); an offset
// >= the descriptor's byte count reads as 0, which is how masked taps / padded rows are fetched (offset 0xFFFFFFFF).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned bload_u32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0);
}
__device__ __forceinline__ float bload_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// 16-byte raw buffer load (4-byte alignment suffices; range checking is per dword at the END of the buffer, but an offset
// that wraps below 0 reads 0 in all four dwords -- tools/ubench/loads.hip)
__device__ __forceinline__ u32x4 bload_u32x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

// out = relu ? max(out, 0) : out;  out = mask ? (mask > 0 ? out : 0) : out -- the epilogue of a split-K launch, whose workgroups
// only hold partial sums (implemented in conv_igemm.hip)
int launch_act_mask(float* out, const float* mask, long long n, int relu, hipStream_t st, int pack = 0);

// name of the main kernel of the most recent convolution entry point (as rocprofv3 prints it), for bench.py's per-kernel
// timing table: ag2v_last_conv_kernel()
void set_last_conv_kernel(const char* name);
void note_chan_sums_done(int v);
bool halo_chan_sums_ok(ConvArgs a);     // conv_halo.hip: this launch would run on a kernel that adds ConvArgs::chan_sums in its epilogue

// K ordering of the packed bf16x3 weight operand of an implicit-GEMM launch over C gathered channels (ConvArgs::tap_major):
// K-tile ((c/32) * R + r) * S + s holds the 32 channels of chunk c/32 at tap (r, s).  One tap per K-tile lets the producers
// address with wave-uniform SGPR offsets.  The taps of a chunk follow each other: the s taps re-read the same cache lines while
// they are still in L1 and the r taps find the neighbouring rows in L2.  Measured alternatives: c fastest over all of C (every
// tap walks all channels) was 6 % slower at C = 512 than the c-major table path; r outermost (r, chunk, s) fetched the
// 8 x 512 x 256 x 256 input 3.3 GB instead of 1.3 GB per launch (FETCH_SIZE: one pass over all channels per filter row, 48 MB per
// XCD between reuses).
static inline bool igemm_tap_major(int C, bool split) {
    static const bool off = false;   // A/B aid: the c-major table path for every shape
    return split && C % 32 == 0 && !off;
}
__host__ __device__ __forceinline__ int tap_major_row(int c, int r, int s, int R, int S) {
    return (((((c >> 5) * R + r) * S + s)) << 5) + (c & 31);
}

// launchers implemented in conv_bf16x3.hip
int launch_igemm_bf16x3(ConvArgs& a, hipStream_t st, bool out_zeroed);
// conv_halo.hip: 3x3 stride-1 launches on split-word inputs as halo tiles staged by LDS-DMA (launch_igemm_bf16x3 asks first)
bool igemm_halo_ok(const ConvArgs& a);
int launch_igemm_halo(ConvArgs& a, hipStream_t st);
void plan_wgrad_bf16x3(ConvArgs& a);      // tiles, pixel-range splits per group (a.spg), a.splitk, a.kchunk -- what the launch will use
int launch_wgrad_bf16x3(ConvArgs& a, hipStream_t st);
// conv_wgrad_halo.hip: 3x3 stride-1 weight gradients on split-word operands, both staged by LDS-DMA (plan_ / launch_wgrad_bf16x3 ask first)
bool wgrad_halo_ok(const ConvArgs& a);
void plan_wgrad_halo(ConvArgs& a);
int launch_wgrad_halo(ConvArgs& a, hipStream_t st);
bool wgrad_f32_ws_ok(const ConvArgs& a);   // exact-fp32 weight gradient on the 16-byte-load pipeline (ConvArgs::f32 = 1, a_bytes / b_bytes set)
// dw[g][i] = (accumulate ? dw[g][i] : 0) + sum_{sl < spg, in order} slabs[(g * spg + sl) * mn + i]   (conv_igemm.hip)
int launch_wgrad_det_reduce(const float* slabs, float* dw, long long mn, int groups, int spg, int accumulate, hipStream_t st);
