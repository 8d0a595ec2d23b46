// Action-graph kernels: GraphTripleConv gather / masked average pooling
// (models/graph_models/graph.py:64-100), the small dense layers of build_mlp / nn.Linear (models/layers.py:6-25,
// graph.py:70-71,103) on the fp32 matrix pipe, and embedding lookups (models/attribute_embed.py:38-45).
//
// The reference loops over the batch in Python and uses boolean-mask indexing (a device->host sync per sample per
// layer).  Here every sample of the batch is handled in one launch: edges are read with coalesced index loads, and the
// pooling is a segmented reduce over the (<= a few dozen) edges of each object -- one workgroup per (sample, object),
// no atomics, deterministic order.
#include <cstdint>
#include <cstdlib>
#include "common.h"

// triples[b,t,:] = [ obj[b, s(b,t), :Din] | pred[b,t,:Dp] | obj[b, o(b,t), :Din] ]     edges: int64 [B,T,2]
__global__ __launch_bounds__(256) void triple_gather_fwd_kernel(const float* __restrict__ obj, const float* __restrict__ pred,
                                                                const long long* __restrict__ edges, float* __restrict__ out,
                                                                int O, int T, int Din, int Dp) {
    const int bt = blockIdx.x;  // b*T + t
    const int b = bt / T;
    const int s = (int)edges[(size_t)bt * 2], o = (int)edges[(size_t)bt * 2 + 1];
    const int Dt = 2 * Din + Dp;
    const float* so = obj + ((size_t)b * O + s) * Din;
    const float* oo = obj + ((size_t)b * O + o) * Din;
    const float* pp = pred + (size_t)bt * Dp;
    float* dst = out + (size_t)bt * Dt;
    for (int i = threadIdx.x; i < Dt; i += 256) {
        float v;
        if (i < Din) v = so[i];
        else if (i < Din + Dp) v = pp[i - Din];
        else v = oo[i - Din - Dp];
        dst[i] = v;
    }
}

// adjoint of the gather: dobj[b,o,:] = sum_{t: s(t)=o} dtri[b,t,:Din] + sum_{t: o(t)=o} dtri[b,t,Din+Dp:], dpred = middle slice.
// grid: (B*O) blocks for dobj; dpred is written by the first O-block of every sample.
__global__ __launch_bounds__(256) void triple_gather_bwd_kernel(const float* __restrict__ dtri, const long long* __restrict__ edges,
                                                                float* __restrict__ dobj, float* __restrict__ dpred, int O, int T,
                                                                int Din, int Dp) {
    const int bo = blockIdx.x;
    const int b = bo / O, o = bo - b * O;
    const int Dt = 2 * Din + Dp;
    for (int i = threadIdx.x; i < Din; i += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t e = (size_t)b * T + t;
            const int s = (int)edges[e * 2], oo = (int)edges[e * 2 + 1];
            if (s == o) acc += dtri[e * Dt + i];
            if (oo == o) acc += dtri[e * Dt + Din + Dp + i];
        }
        dobj[(size_t)bo * Din + i] = acc;
    }
    if (o == 0 && dpred) {
        for (int t = 0; t < T; ++t) {
            const size_t e = (size_t)b * T + t;
            for (int i = threadIdx.x; i < Dp; i += 256) dpred[e * Dp + i] = dtri[e * Dt + Din + i];
        }
    }
}

// masked average pooling: pooled[b,o,:Hd] = (sum_{valid t, s(t)=o} newt[b,t,:Hd] + sum_{valid t, o(t)=o} newt[b,t,Hd+Dp:]) / count
// count = number of valid incident edge endpoints (graph.py:94-99); rows with count 0 stay 0.  valid: uint8 [B,T].
__global__ __launch_bounds__(256) void triple_pool_fwd_kernel(const float* __restrict__ newt, const long long* __restrict__ edges,
                                                              const unsigned char* __restrict__ valid, float* __restrict__ pooled,
                                                              float* __restrict__ counts, int O, int T, int Hd, int Dp) {
    const int bo = blockIdx.x;
    const int b = bo / O, o = bo - b * O;
    const int Dt = 2 * Hd + Dp;
    int cnt = 0;
    for (int t = 0; t < T; ++t) {
        const size_t e = (size_t)b * T + t;
        if (!valid[e]) continue;
        cnt += ((int)edges[e * 2] == o) + ((int)edges[e * 2 + 1] == o);
    }
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    for (int i = threadIdx.x; i < Hd; i += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const size_t e = (size_t)b * T + t;
            if (!valid[e]) continue;
            if ((int)edges[e * 2] == o) acc += newt[e * Dt + i];
            if ((int)edges[e * 2 + 1] == o) acc += newt[e * Dt + Hd + Dp + i];
        }
        pooled[(size_t)bo * Hd + i] = cnt > 0 ? acc / (float)cnt : acc;
    }
    (void)inv;
    if (threadIdx.x == 0 && counts) counts[bo] = (float)cnt;
}

// adjoint: dnewt[b,t,:Hd] = valid ? dpooled[b,s(t)]/count[s(t)] : 0 ; dnewt[b,t,Hd+Dp:] likewise for o(t); the predicate slice
// dnewt[b,t,Hd:Hd+Dp] receives dp_in (gradient of the returned new_p vectors) or 0.
__global__ __launch_bounds__(256) void triple_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ dp_in,
                                                              const long long* __restrict__ edges,
                                                              const unsigned char* __restrict__ valid, const float* __restrict__ counts,
                                                              float* __restrict__ dnewt, int O, int T, int Hd, int Dp) {
    const int bt = blockIdx.x;
    const int b = bt / T;
    const int Dt = 2 * Hd + Dp;
    const int s = (int)edges[(size_t)bt * 2], o = (int)edges[(size_t)bt * 2 + 1];
    const bool v = valid[bt] != 0;
    const float cs = counts[(size_t)b * O + s], co = counts[(size_t)b * O + o];
    const float is = (v && cs > 0.f) ? 1.0f / cs : 0.f, io = (v && co > 0.f) ? 1.0f / co : 0.f;
    const float* ds = dpooled + ((size_t)b * O + s) * Hd;
    const float* dov = dpooled + ((size_t)b * O + o) * Hd;
    float* dst = dnewt + (size_t)bt * Dt;
    for (int i = threadIdx.x; i < Dt; i += 256) {
        float g;
        if (i < Hd) g = v ? ds[i] * is : 0.f;
        else if (i < Hd + Dp) g = dp_in ? dp_in[(size_t)bt * Dp + (i - Hd)] : 0.f;
        else g = v ? dov[i - Hd - Dp] * io : 0.f;
        dst[i] = g;
    }
}

// ---------------------------------------------------------------- small dense GEMM on v_mfma_f32_32x32x2_f32
// C[M,N] = act( A * B + bias ),  A[m,k] = a[m*sam + k*sak],  B[k,n] = b[k*sbk + n*sbn],  C row-major (ldc).
// 64x64 tile per workgroup; KS groups of 4 waves (one 32x32 MFMA tile per wave) split the K range of the tile between them
// and are summed through LDS: the graph nets' GEMMs have 1-11 row tiles and K up to 1152, so the serial K loop, not the
// chip, bounds a launch (25 us at KS = 1).  BK = 16.
template <int KS>
__global__ __launch_bounds__(256 * KS) void gemm_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ bias, float* __restrict__ c, int M, int N,
                                                            int K, long long sam, long long sak, long long sbk, long long sbn,
                                                            int ldc, int relu, int accumulate) {
    // BK = 32 with four K groups: the loop is a chain of (global-load latency, barrier) steps -- ~0.8 us each against 0.25 us of MFMAs
    // -- so halving its trip count is worth more than the LDS it costs (139 KB: these launches have <= 24 workgroups anyway); the
    // cross-group reduction buffer aliases the A image, which is dead by then.
    constexpr int BK = (KS > 1) ? 32 : 16, LD = 64 + 4, EL = BK / 4;    // EL elements of each operand per thread and K-tile
    __shared__ float As[KS][2][BK][LD];
    __shared__ float Bs[KS][2][BK][LD];
    float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(&As[0][0][0][0]);
    static_assert(KS == 1 || sizeof(float) * (KS - 1) * 4 * 16 * 64 <= sizeof(float) * KS * 2 * BK * LD, "reduction buffer fits the A image");
    const int kg = threadIdx.x >> 8;            // K group
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk_all = (K + BK - 1) / BK;
    const int nk = (nk_all + KS - 1) / KS;      // K-tiles per group (the same trip count for every group: uniform barriers)
    const int kbase = kg * nk * BK;
    const int kend = min(K, kbase + nk * BK);
    // Each thread stages 4 A and 4 B elements per K-tile.  Lanes run along whichever index is contiguous in memory:
    // k-contiguous operand -> (kk = tid & 15, row = (tid >> 4) + 16 i); otherwise (row = tid & 63, kk = (tid >> 6) + 4 i).
    const bool a_kc = (sak == 1), b_kc = (sbk == 1);
    // thread -> (k, row) of element i: k-contiguous operand: kk = (tid & 15) + 16 (i / 4), row = (tid >> 4) + 16 (i % 4);
    // row-contiguous operand: row = tid & 63, kk = (tid >> 6) + 4 i
    float ra[EL], rb[EL];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < EL; ++i) {
            const int akk = a_kc ? ((tid & 15) + 16 * (i >> 2)) : ((tid >> 6) + 4 * i), arow = a_kc ? ((tid >> 4) + 16 * (i & 3)) : (tid & 63);
            const int bkk = b_kc ? ((tid & 15) + 16 * (i >> 2)) : ((tid >> 6) + 4 * i), brow = b_kc ? ((tid >> 4) + 16 * (i & 3)) : (tid & 63);
            const int m = m0 + arow, n = n0 + brow;
            ra[i] = (m < M && k0 + akk < kend) ? a[(long long)m * sam + (long long)(k0 + akk) * sak] : 0.f;
            rb[i] = (n < N && k0 + bkk < kend) ? b[(long long)(k0 + bkk) * sbk + (long long)n * sbn] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < EL; ++i) {
            const int akk = a_kc ? ((tid & 15) + 16 * (i >> 2)) : ((tid >> 6) + 4 * i), arow = a_kc ? ((tid >> 4) + 16 * (i & 3)) : (tid & 63);
            const int bkk = b_kc ? ((tid & 15) + 16 * (i >> 2)) : ((tid >> 6) + 4 * i), brow = b_kc ? ((tid >> 4) + 16 * (i & 3)) : (tid & 63);
            As[kg][buf][akk][arow] = ra[i];
            Bs[kg][buf][bkk][brow] = rb[i];
        }
    };
    fetch(kbase);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) fetch(kbase + (kt + 1) * BK);        // next tile's loads are in flight during this tile's MFMAs
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kg][cur][kk][wm0 + (lane & 31)], Bs[kg][cur][kk][wn0 + (lane & 31)], acc, 0,
                                                       0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (KS > 1) {
        // (the loop's last barrier is behind every group's last read of the A image that `red` aliases)
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[kg - 1][wave][r][lane] = acc[r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 0; g < KS - 1; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[g][wave][r][lane];
    }
    const int n = n0 + wn0 + (lane & 31);
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
        // the 16 read-backs of an accumulating launch are requested together (one conditional load per register, each followed
        // by s_waitcnt vmcnt(0), serialised 16 memory round trips behind the stores of the previous registers)
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
        if (accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < M) old[r] = c[(size_t)m * ldc + n];
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        const float floor_ = relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < 16; ++r) {                                                 // straight-line: no use of a loaded word
            old[r] = fmaxf(acc[r] + bv + old[r], floor_);
            asm volatile("" : "+v"(old[r]));      // (keeps the compiler from sinking the arithmetic into the store blocks)
        }
#pragma unroll                                                                         // inside a conditional block
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) c[(size_t)m * ldc + n] = old[r];
        }
    }
}

// ---------------------------------------------------------------- one-round-trip GEMM for the graph nets' shapes
// The graph nets run 15 dependent time steps of ~13 GEMMs with 48-128 rows and K = 128-1152: a launch is a latency chain, not
// work (gemm_f32_kernel<4> above: 18-21 us for 42 MFLOP -- launch, then 4 x (global load -> LDS -> barrier -> MFMA), cross-group
// sum, epilogue).  Here a workgroup owns a 32 x 32 output tile and its 16 waves split K between them; every wave requests its
// whole K slice at once, straight into the registers the MFMAs read (the order of the k terms inside a sum is free, so lane
// (i, g) simply takes k = 8 q + 4 g + e of the slice for BOTH operands: 16-byte loads along k for a k-contiguous operand,
// coalesced 4-byte loads for a row-contiguous one, no LDS staging).  One memory round trip, <= 36 MFMAs, one LDS reduction
// over the 16 waves, epilogue.  Requires K % 4 == 0 (16-byte loads; every K on the path).
template <int CHQ>
__global__ __launch_bounds__(1024) void gemm_f32_oneshot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                const float* __restrict__ bias, float* __restrict__ c, int M, int N,
                                                                int K, long long sam, long long sak, long long sbk, long long sbn,
                                                                int ldc, int relu, int accumulate, int slice, int avec, int bvec) {
    __shared__ float red[16][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, i32 = lane & 31, g = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kbeg = wv * slice, kend = min(K, kbeg + slice);
    const int m = m0 + i32, n = n0 + i32;
    const bool a_kc = avec != 0, b_kc = bvec != 0;       // k-contiguous AND 16-byte aligned rows: quad loads
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = kbeg; kc < kend; kc += 8 * CHQ) {
        float ra[4 * CHQ], rb[4 * CHQ];
#pragma unroll
        for (int q = 0; q < CHQ; ++q) {
            const int k = kc + 8 * q + 4 * g;
            const bool kin = k < kend;          // (k and kend are multiples of 4: a quad is in or out as a whole)
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (kin && m < M) {
                if (a_kc) va = *reinterpret_cast<const float4*>(a + (long long)m * sam + k);
                else {
                    const float* ap = a + (long long)m * sam + (long long)k * sak;
                    va = make_float4(ap[0], ap[sak], ap[2 * sak], ap[3 * sak]);
                }
            }
            if (kin && n < N) {
                if (b_kc) vb = *reinterpret_cast<const float4*>(b + (long long)n * sbn + k);
                else {
                    const float* bp = b + (long long)n * sbn + (long long)k * sbk;
                    vb = make_float4(bp[0], bp[sbk], bp[2 * sbk], bp[3 * sbk]);
                }
            }
            ra[4 * q] = va.x; ra[4 * q + 1] = va.y; ra[4 * q + 2] = va.z; ra[4 * q + 3] = va.w;
            rb[4 * q] = vb.x; rb[4 * q + 1] = vb.y; rb[4 * q + 2] = vb.z; rb[4 * q + 3] = vb.w;
        }
#pragma unroll
        for (int j = 0; j < 4 * CHQ; ++j)
            if (kc + 8 * (j >> 2) < kend)        // wave-uniform: whole 8-k groups beyond the slice hold zeros only
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[j], rb[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r * 64 + lane] = acc[r];
    __syncthreads();
    const int t = threadIdx.x;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) sum += red[w][t];
    const int r = t >> 6, l2 = t & 63;
    const int mo = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l2 >> 5), no = n0 + (l2 & 31);
    if (mo < M && no < N) {
        float v = sum + (bias ? bias[no] : 0.f);
        if (accumulate) v += c[(size_t)mo * ldc + no];
        c[(size_t)mo * ldc + no] = relu ? fmaxf(v, 0.f) : v;
    }
}

// g = dy * (y > 0)   (ReLU backward), and column sums for bias gradients
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ g,
                                                       long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += x[(size_t)m * N + n];
    out[n] = s;
}

// ---------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ table,
                                                            float* __restrict__ out, int D, int out_stride, int out_off) {
    const long long r = idx[blockIdx.x];
    for (int i = threadIdx.x; i < D; i += 256) out[(size_t)blockIdx.x * out_stride + out_off + i] = table[(size_t)r * D + i];
}
// dtable[v,:] = sum_{i: idx[i]=v} dout[i,:]  -- one workgroup per vocabulary row (tables have <= a few dozen rows)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, int n, int D, int in_stride, int in_off) {
    const long long v = blockIdx.x;
    for (int i = threadIdx.x; i < D; i += 256) {
        float acc = 0.f;
        for (int r = 0; r < n; ++r)
            if (idx[r] == v) acc += dout[(size_t)r * in_stride + in_off + i];
        dtable[(size_t)v * D + i] = acc;
    }
}

// ================================================================ C ABI
extern "C" int ag2v_triple_gather_fwd(const float* obj, const float* pred, const long long* edges, float* out, int B, int O, int T,
                                      int Din, int Dp, void* stream) {
    if (!obj || !pred || !edges || !out || B <= 0 || O <= 0 || T <= 0 || Din <= 0 || Dp <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(triple_gather_fwd_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, obj, pred, edges, out, O, T, Din, Dp);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_triple_gather_bwd(const float* dtri, const long long* edges, float* dobj, float* dpred, int B, int O, int T,
                                      int Din, int Dp, void* stream) {
    if (!dtri || !edges || !dobj || B <= 0 || O <= 0 || T <= 0 || Din <= 0 || Dp <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(triple_gather_bwd_kernel, dim3(B * O), dim3(256), 0, (hipStream_t)stream, dtri, edges, dobj, dpred, O, T, Din, Dp);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_triple_pool_fwd(const float* newt, const long long* edges, const unsigned char* valid, float* pooled, float* counts,
                                    int B, int O, int T, int Hd, int Dp, void* stream) {
    if (!newt || !edges || !valid || !pooled || !counts || B <= 0 || O <= 0 || T <= 0 || Hd <= 0 || Dp < 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(triple_pool_fwd_kernel, dim3(B * O), dim3(256), 0, (hipStream_t)stream, newt, edges, valid, pooled, counts, O, T, Hd, Dp);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_triple_pool_bwd(const float* dpooled, const float* dp_in, const long long* edges, const unsigned char* valid,
                                    const float* counts, float* dnewt, int B, int O, int T, int Hd, int Dp, void* stream) {
    if (!dpooled || !edges || !valid || !counts || !dnewt || B <= 0 || O <= 0 || T <= 0 || Hd <= 0 || Dp < 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(triple_pool_bwd_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, dpooled, dp_in, edges, valid, counts, dnewt, O, T, Hd, Dp);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// Generic strided GEMM (see gemm_f32_kernel).  The three uses of a Linear layer y = x W^T + b (x[M,K], W[N,K]):
//   forward   C=y [M,N]:  A=x (sam=K,sak=1),  B=W^T (sbk=1,sbn=K), bias, relu
//   dgrad     C=dx[M,K]:  A=g (sam=N,sak=1),  B=W   (sbk=K,sbn=1)
//   wgrad     C=dW[N,K]:  A=g^T (sam=1,sak=N), B=x  (sbk=K,sbn=1)
extern "C" int ag2v_gemm_f32(const float* a, const float* b, const float* bias, float* c, int M, int N, int K, long long sam,
                             long long sak, long long sbk, long long sbn, int ldc, int relu, int accumulate, void* stream) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || ldc < N) return AG2V_EINVAL;
    constexpr int oneshot_mode = 1;
    const long long tiles32 = (long long)ceil_div(N, 32) * ceil_div(M, 32);
    if (oneshot_mode && K % 4 == 0 && K <= 16 * 72 && tiles32 <= 1024) {
        const int slice = ((ceil_div(K, 16) + 7) / 8) * 8;
        const dim3 g32(ceil_div(N, 32), ceil_div(M, 32));
        // quad loads need 16-byte aligned rows (parameters are views into the optimizers' flat buffers: any 4-byte offset occurs)
        const int avec = sak == 1 && sam % 4 == 0 && ((uintptr_t)a & 15) == 0, bvec = sbk == 1 && sbn % 4 == 0 && ((uintptr_t)b & 15) == 0;
        if (slice <= 32)
            hipLaunchKernelGGL(gemm_f32_oneshot_kernel<4>, g32, dim3(1024), 0, (hipStream_t)stream, a, b, bias, c, M, N, K, sam, sak, sbk,
                               sbn, ldc, relu, accumulate, slice, avec, bvec);
        else
            hipLaunchKernelGGL(gemm_f32_oneshot_kernel<9>, g32, dim3(1024), 0, (hipStream_t)stream, a, b, bias, c, M, N, K, sam, sak, sbk,
                               sbn, ldc, relu, accumulate, slice, avec, bvec);
        AG2V_CHECK_LAUNCH();
        return AG2V_OK;
    }
    const dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
    if (K >= 256 && (long long)grid.x * grid.y < 512)   // few tiles, long K: four K groups per tile
        hipLaunchKernelGGL(gemm_f32_kernel<4>, grid, dim3(1024), 0, (hipStream_t)stream, a, b, bias, c, M, N, K, sam, sak, sbk, sbn, ldc,
                           relu, accumulate);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, b, bias, c, M, N, K, sam, sak, sbk, sbn, ldc,
                           relu, accumulate);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_relu_bwd(const float* dy, const float* y, float* g, long long n, void* stream) {
    if (!dy || !y || !g || n <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, g, n);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_col_sum(const float* x, float* out, int M, int N, void* stream) {
    if (!x || !out || M <= 0 || N <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(col_sum_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, x, out, M, N);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_embedding_fwd(const long long* idx, const float* table, float* out, int n, int D, int out_stride, int out_off,
                                  void* stream) {
    if (!idx || !table || !out || n <= 0 || D <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, idx, table, out, D, out_stride, out_off);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_embedding_bwd(const long long* idx, const float* dout, float* dtable, int n, int V, int D, int in_stride,
                                  int in_off, void* stream) {
    if (!idx || !dout || !dtable || n <= 0 || V <= 0 || D <= 0) return AG2V_EINVAL;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, idx, dout, dtable, n, D, in_stride, in_off);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
