// Graph-net PROGRAMS: a whole recurrent pass of an action-graph net as ONE persistent launch.
//
// The graph nets (reference: models/graph_models/model.py:126-169 Acts2LayoutModel's loop over time steps, graph.py:41-107
// GraphTripleConv, discriminator.py:273-313 get_obj_vecs) are chains of ~20 DEPENDENT small operators per time step -- GEMMs with
// 88-136 rows, gathers over <= 16 edges per sample, masked average pooling -- times 15 time steps, forward and backward, for three
// nets: ~3 400 launches of 3-12 us per training step (0.1 % of the FLOPs, ~30 ms of the step), each one bounded by launch + one memory
// round trip, and too register-hungry neighbours (the convolution tiles fill every SIMD's VGPR file) for a second stream to hide them.
//
// Here the host writes the pass as a PROGRAM -- an array of operator descriptors (GOp) over tensors of one pre-planned arena -- and a
// single launch of G co-resident workgroups interprets it: every workgroup takes its share of an operator's output tiles / rows, then
// all meet at a grid barrier.  No tensor is reused inside a program (a forward value is the backward's mask), so the only hazards are
// producer -> consumer across workgroups on different XCDs (private, non-coherent L2s).  Visibility without cache-wide fences -- an
// agent-scope release / acquire per barrier writes back and invalidates whole L2s: 21 us per operator measured, the weights re-fetched
// from memory for every operator -- rests on a WRITE-ONCE discipline (cdna_hip_programming.md, Guideline 16, sc1 payload form):
//   * every store of the program is WRITE-THROUGH (sc1: __hip_atomic_store, relaxed, agent scope) and drained (s_waitcnt vmcnt(0))
//     by its wave before the workgroup arrives at the barrier, so the bytes are in memory when the arrival counter moves;
//   * every tensor is written by exactly one operator and read only by later ones, and tensors never share a 128-byte line (the
//     arena hands out 128-byte-aligned blocks): no cache can hold a line of a tensor from before it was written, so plain loads
//     after the barrier miss and fetch the written bytes (caches start clean at the launch boundary; nothing prefetches);
//   * the arrival counter is only touched by agent-scope atomics.
//
// Operators (all fp32; matrices row-major with explicit row strides, so slices and stacks are views):
//   GEMM        C[m,n] = act(sum_k A'[m,k] B[n,k] + bias[n]) + res[m,n],  A' = A, or (mask > 0 ? A : 0) -- the backward of a ReLU
//               whose forward output is `mask`; exact fp32 MFMA (v_mfma_f32_32x32x2_f32), one 32 x 32 output tile per workgroup
//               pass, the reduction split over the 16 waves, every wave's whole K slice requested at once (16-byte loads along k).
//   CAT         out[r,:] = [seg0[i0(r)] | seg1[i1(r)] | seg2[i2(r)]]: concatenation of up to three row-gathered segments
//               (graph.py:64-70 builds [obj[s] | pred | obj[o]]; model.py:128 cat(emb, boxes)).
//   POOL        masked average pooling of the subject / object parts of the new triple vectors onto objects (graph.py:79-100);
//   POOL_BWD    its adjoint (+ the gradient of the predicate part);   GATHER_BWD  the adjoint of the gather in CAT for obj rows;
//   TRANSPOSE   out[k,n] = in[n,k] (the backward GEMMs read W^T so that their reduction index is contiguous);
//   ADD3        c = a + b + d (recurrent box gradient).
#include "common.h"

enum { GOP_GEMM = 1, GOP_CAT = 2, GOP_POOL = 3, GOP_POOL_BWD = 4, GOP_GATHER_BWD = 5, GOP_TRANSPOSE = 6, GOP_ADD3 = 7 };

struct GOp {                // 192 bytes (ag2v_graph_op_bytes()); filled by the host (ag2video_amd/graph_program.py)
    int type, M, N, K;
    const float* a;         // GEMM: A [M, K] | CAT: segment 0 | POOL: newt | POOL_BWD: dpooled | GATHER_BWD: dtri | TRANSPOSE / ADD3: in
    const float* b;         // GEMM: B [N, K] | CAT: segment 1 | POOL_BWD: dp_in (nullable) | ADD3: second input (nullable)
    const float* d;         // CAT: segment 2 | ADD3: third input (nullable)
    const float* bias;      // GEMM: [N] (nullable)
    const float* res;       // GEMM: residual [M, N] (nullable)
    const float* mask;      // GEMM: ReLU mask for A (nullable) | POOL_BWD / POOL: counts (float, [B * O])
    float* c;               // output
    const int* i0;          // CAT: row index of segment 0 (nullable = identity) | POOL*: subject object row (b * O + s), [B * T]
    const int* i1;          // CAT: segment 1 index | POOL*: object object row (b * O + o)
    const int* i2;          // CAT: segment 2 index
    const unsigned char* valid;   // POOL / POOL_BWD: edge validity [B * T]
    long long lda, ldb, ldd, ldr, ldm, ldc;
    int w0, w1, w2;         // CAT: segment widths | POOL*: w0 = Hd, w1 = Dp | GATHER_BWD: w0 = Din, w1 = Dp
    int relu, acc, O, T;    // GEMM: relu / accumulate into c | graph ops: objects, edges per sample
    int pad_[3];
};
static_assert(sizeof(GOp) == 192, "GOp layout is part of the host interface");

#define GP_MAX_EDGES 64              // edges per sample a POOL / GATHER_BWD operator can take (one 64-bit incidence mask per endpoint)
#define GP_THREADS 1024
#define GP_WAVES 16
// A grid that is not co-resident ends with the timeout flag set instead of hanging.  The limit is wall time (s_memrealtime, 100 MHz):
// 2 s per barrier -- workgroups that find their CUs taken by another stream's kernels (a persistent convolution, RCCL's all-reduce on
// the communication stream at > 1 rank) start when those retire, tens of ms later at most (tests/test_graph_program_gpu.py holds 192
// of the 256 CUs for 100 ms beside a run); only a grid that can NEVER be resident reaches the limit.
#define GP_BARRIER_TIMEOUT_TICKS 200000000ull

// write-through store (sc1): the value is in memory once the storing wave's vmcnt has drained
__device__ __forceinline__ void st_wt(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- grid barrier (arrival counter; visibility: see the header)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, unsigned* timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        unsigned long long t0 = 0;
        // a barrier of this run has already given up (this workgroup's or another's): the results are invalid anyway -- do not wait
        // up to the limit again at each of the remaining n_ops barriers before the host sees the flag
        bool dead = __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        while (!dead && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {          // look at the clock (and at the flag) every ~1 k polls
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > GP_BARRIER_TIMEOUT_TICKS) { __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                dead = __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------- GEMM
// one 32 x 32 tile: wave w takes k in [w * Kw, (w + 1) * Kw), Kw a multiple of 8; lane (r = lane & 31, h = lane >> 5) loads
// A[m0 + r][k8 + 4 h .. + 3] and B[n0 + r][k8 + 4 h .. + 3] for every k8 of its slice at once; MFMA j of a k8 step multiplies the
// k = k8 + j (h = 0) and k = k8 + 4 + j (h = 1) columns.  Partials meet in LDS [16][1024].
typedef unsigned gp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gp_rsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes > 0xFFFFFFF0LL ? 0xFFFFFFF0u : (unsigned)bytes, 0x00020000);
}
// One 32 x 32 output tile.  Wave w owns the k slice [w Kw, (w + 1) Kw), Kw a multiple of 8, and walks it in chunks of 32 columns:
//   * the chunk of A (and of the ReLU mask) and of B is loaded COALESCED -- eight lanes cover the 128 contiguous bytes of one row,
//     a wave instruction eight whole rows (a first version loaded MFMA-fragment-shaped, one 16-byte piece of each of 32 rows per
//     instruction: 64 partial cache lines per wave-load x 16 waves kept the CU's address unit busy for ~8 us per tile);
//   * staged through a wave-private LDS block [32][33] per operand (only this wave reads what it wrote: no workgroup barrier),
//     from which the MFMA operands are conflict-free ds_read_b32; the next chunk's loads are in flight meanwhile;
//   * the 16 partial tiles meet in LDS (aliasing the staging blocks) and the 1024 threads finish one element each.
#define GP_CH 32
#define GP_PITCH 33
__device__ __forceinline__ void gemm_tile(const GOp& o, int tm, int tn, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = o.K;                                          // K % 4 == 0, every leading dimension % 4 == 0, 16-byte aligned bases
    const int Kw = ((K + GP_WAVES * 8 - 1) / (GP_WAVES * 8)) * 8;
    const int k_lo = wave * Kw, k_hi = min(K, k_lo + Kw);
    const __amdgpu_buffer_rsrc_t ars = gp_rsrc(o.a, ((long long)(o.M - 1) * o.lda + K) * 4);
    const __amdgpu_buffer_rsrc_t brs = gp_rsrc(o.b, ((long long)(o.N - 1) * o.ldb + K) * 4);
    const bool masked = o.mask != nullptr;
    const __amdgpu_buffer_rsrc_t mrs = gp_rsrc(masked ? o.mask : o.a, ((long long)(o.M - 1) * (masked ? o.ldm : o.lda) + K) * 4);
    float* sa = lds + wave * (2 * 32 * GP_PITCH);               // this wave's staging blocks
    float* sb = sa + 32 * GP_PITCH;
    // unit u = it * 64 + lane of a chunk: row u >> 3, four columns 4 (u & 7)
    const int urow = lane >> 3, ucol = 4 * (lane & 7);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    gp_u32x4 ra[4], rb[4], rm[4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + urow, k = kc + ucol;
            const bool kok = k < k_hi;
            const int m = tm * 32 + row, n = tn * 32 + row;
            const unsigned ao = (kok && m < o.M) ? (unsigned)(((long long)m * o.lda + k) * 4) : 0xFFFFFFFFu;
            const unsigned bo = (kok && n < o.N) ? (unsigned)(((long long)n * o.ldb + k) * 4) : 0xFFFFFFFFu;
            ra[it] = __builtin_amdgcn_raw_buffer_load_b128(ars, ao, 0, 0);
            rb[it] = __builtin_amdgcn_raw_buffer_load_b128(brs, bo, 0, 0);
            if (masked) {
                const unsigned mo = (kok && m < o.M) ? (unsigned)(((long long)m * o.ldm + k) * 4) : 0xFFFFFFFFu;
                rm[it] = __builtin_amdgcn_raw_buffer_load_b128(mrs, mo, 0, 0);
            }
        }
    };
    const int r = lane & 31, h = lane >> 5;
    if (k_lo < k_hi) fetch(k_lo);
    for (int kc = k_lo; kc < k_hi; kc += GP_CH) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + urow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = __uint_as_float(ra[it][j]);
                if (masked) a = (__uint_as_float(rm[it][j]) > 0.f) ? a : 0.f;
                sa[row * GP_PITCH + ucol + j] = a;
                sb[row * GP_PITCH + ucol + j] = __uint_as_float(rb[it][j]);
            }
        }
        if (kc + GP_CH < k_hi) fetch(kc + GP_CH);               // next chunk in flight under this chunk's MFMAs
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave's LDS writes have landed (wave-private block)
        const int nk = min(GP_CH, k_hi - kc);
#pragma unroll 4
        for (int k2 = 0; k2 < nk; k2 += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[r * GP_PITCH + k2 + h], sb[r * GP_PITCH + k2 + h], acc, 0, 0, 0);
    }
    __syncthreads();                                   // every wave is done with its staging block: the reduction buffer aliases them
    float* red = lds;
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave * 1024 + i * 64 + lane] = acc[i];
    __syncthreads();
    const int e = threadIdx.x;                          // element e = i * 64 + l of the tile
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < GP_WAVES; ++w) s += red[w * 1024 + e];
    const int i = e >> 6, l = e & 63;
    const int row = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = tn * 32 + (l & 31);
    if (row < o.M && col < o.N) {
        if (o.bias) s += o.bias[col];
        if (o.relu) s = fmaxf(s, 0.f);
        if (o.res) s += o.res[(size_t)row * o.ldr + col];
        float* dst = o.c + (size_t)row * o.ldc + col;
        st_wt(dst, o.acc ? *dst + s : s);
    }
    __syncthreads();                                   // the reduction reads are done before the next tile stages into the same LDS
}

// ---------------------------------------------------------------- element-parallel operators (gid = one group of 4 columns of one row)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st_wt4(float* p, float4 v) { st_wt(p, v.x); st_wt(p + 1, v.y); st_wt(p + 2, v.z); st_wt(p + 3, v.w); }

// every segment width and leading dimension is a multiple of 4 (builder)
__device__ __forceinline__ void op_cat(const GOp& o, long long gid, long long nthreads) {
    const int W4 = (o.w0 + o.w1 + o.w2) >> 2;
    for (long long u = gid; u < (long long)o.M * W4; u += nthreads) {
        const int row = (int)(u / W4), c = 4 * (int)(u - (long long)row * W4);
        const float* src;
        if (c < o.w0) src = o.a + (size_t)(o.i0 ? o.i0[row] : row) * o.lda + c;
        else if (c < o.w0 + o.w1) src = o.b + (size_t)(o.i1 ? o.i1[row] : row) * o.ldb + (c - o.w0);
        else src = o.d + (size_t)(o.i2 ? o.i2[row] : row) * o.ldd + (c - o.w0 - o.w1);
        st_wt4(o.c + (size_t)row * o.ldc + c, ld4(src));
    }
}
// pooled[bo, :Hd] = (sum over valid edges t of sample b with s(t) == bo of newt[t, :Hd] + ... o(t) == bo of newt[t, Hd + Dp:]) / count
// T <= GP_MAX_EDGES (64; the host checks ag2v_graph_program_max_edges() before it builds a program, a larger T takes the
// per-operator path): the incidence of the sample's edges on this object is gathered into two bit masks first (independent index loads), then the
// matching rows are summed (independent data loads) -- a loop of "load index, compare, load data" was a 17-deep dependent chain (51 us)
__device__ __forceinline__ void edge_masks(const GOp& o, int b, int bo, bool use_valid, unsigned long long& ms, unsigned long long& mo) {
    ms = mo = 0;
#pragma unroll 8
    for (int t = 0; t < o.T; ++t) {
        const int e = b * o.T + t;
        const bool v = !use_valid || o.valid[e] != 0;
        ms |= (v && o.i0[e] == bo) ? (1ull << t) : 0ull;
        mo |= (v && o.i1[e] == bo) ? (1ull << t) : 0ull;
    }
}
__device__ __forceinline__ void op_pool(const GOp& o, long long gid, long long nthreads) {
    const int Hd = o.w0, Dp = o.w1, H4 = Hd >> 2;
    for (long long u = gid; u < (long long)o.M * H4; u += nthreads) {
        const int bo = (int)(u / H4), c = 4 * (int)(u - (long long)bo * H4);
        const int b = bo / o.O;
        unsigned long long ms, mo;
        edge_masks(o, b, bo, true, ms, mo);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int t = 0; t < o.T; ++t) {
            const float* rowp = o.a + (size_t)(b * o.T + t) * o.lda + c;
            if ((ms >> t) & 1ull) { const float4 v = ld4(rowp); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            if ((mo >> t) & 1ull) { const float4 v = ld4(rowp + Hd + Dp); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        const int cnt = __builtin_popcountll(ms) + __builtin_popcountll(mo);
        if (cnt > 0) { const float q = (float)cnt; acc.x /= q; acc.y /= q; acc.z /= q; acc.w /= q; }
        st_wt4(o.c + (size_t)bo * o.ldc + c, acc);
        if (c == 0) st_wt(const_cast<float*>(o.mask) + bo, (float)cnt);      // counts (read by POOL_BWD of the backward program)
    }
}
__device__ __forceinline__ void op_pool_bwd(const GOp& o, long long gid, long long nthreads) {
    const int Hd = o.w0, Dp = o.w1, W4 = (2 * Hd + Dp) >> 2;
    for (long long u = gid; u < (long long)o.M * W4; u += nthreads) {
        const int bt = (int)(u / W4), c = 4 * (int)(u - (long long)bt * W4);
        const bool v = o.valid[bt] != 0;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < Hd || c >= Hd + Dp) {
            const int ob = (c < Hd) ? o.i0[bt] : o.i1[bt];
            const float cn = o.mask[ob];
            const float inv = (v && cn > 0.f) ? 1.0f / cn : 0.f;
            const float4 d = ld4(o.a + (size_t)ob * o.lda + (c < Hd ? c : c - Hd - Dp));
            g = make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv);
        } else if (o.b) {
            g = ld4(o.b + (size_t)bt * o.ldb + (c - Hd));
        }
        st_wt4(o.c + (size_t)bt * o.ldc + c, g);
    }
}
// dobj[bo, :Din] = res[bo] + sum_t (s(t) == bo) dtri[t, :Din] + (o(t) == bo) dtri[t, Din + Dp:]   (every edge: the gather is unmasked)
__device__ __forceinline__ void op_gather_bwd(const GOp& o, long long gid, long long nthreads) {
    const int Din = o.w0, Dp = o.w1, D4 = Din >> 2;
    for (long long u = gid; u < (long long)o.M * D4; u += nthreads) {
        const int bo = (int)(u / D4), c = 4 * (int)(u - (long long)bo * D4);
        const int b = bo / o.O;
        unsigned long long ms, mo;
        edge_masks(o, b, bo, false, ms, mo);
        float4 acc = o.res ? ld4(o.res + (size_t)bo * o.ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int t = 0; t < o.T; ++t) {
            const float* rowp = o.a + (size_t)(b * o.T + t) * o.lda + c;
            if ((ms >> t) & 1ull) { const float4 v = ld4(rowp); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            if ((mo >> t) & 1ull) { const float4 v = ld4(rowp + Din + Dp); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        st_wt4(o.c + (size_t)bo * o.ldc + c, acc);
    }
}
__device__ __forceinline__ void op_add3(const GOp& o, long long gid, long long nthreads) {
    const int N4 = o.N >> 2;
    for (long long u = gid; u < (long long)o.M * N4; u += nthreads) {
        const int row = (int)(u / N4), c = 4 * (int)(u - (long long)row * N4);
        float4 v = ld4(o.a + (size_t)row * o.lda + c);
        if (o.b) { const float4 t = ld4(o.b + (size_t)row * o.ldb + c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (o.d) { const float4 t = ld4(o.d + (size_t)row * o.ldd + c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        st_wt4(o.c + (size_t)row * o.ldc + c, v);
    }
}

__global__ __launch_bounds__(GP_THREADS) void graph_program_kernel(const GOp* __restrict__ ops, int n_ops, unsigned* __restrict__ sync) {
    __shared__ __attribute__((aligned(16))) float red[GP_WAVES * 2 * 32 * GP_PITCH];      // staging blocks / reduction buffer (>= 16 x 1024)
    __shared__ GOp desc[2];
    const int G = gridDim.x, wg = blockIdx.x;
    const long long gid = (long long)wg * GP_THREADS + threadIdx.x, nthreads = (long long)G * GP_THREADS;
    // descriptors through LDS, one copy per workgroup, then uniform reads; the array is constant for the launch, so operator k + 1's
    // record is fetched while operator k runs (it used to cost a memory round trip behind every barrier)
    if (threadIdx.x < (int)(sizeof(GOp) / 4)) reinterpret_cast<unsigned*>(&desc[0])[threadIdx.x] = reinterpret_cast<const unsigned*>(ops)[threadIdx.x];
    __syncthreads();
    for (int k = 0; k < n_ops; ++k) {
        if (k + 1 < n_ops && threadIdx.x < (int)(sizeof(GOp) / 4))
            reinterpret_cast<unsigned*>(&desc[(k + 1) & 1])[threadIdx.x] = reinterpret_cast<const unsigned*>(ops + k + 1)[threadIdx.x];
        const GOp& o = desc[k & 1];
        switch (o.type) {
            case GOP_GEMM: {
                const int tms = (o.M + 31) >> 5, tns = (o.N + 31) >> 5;
                for (int t = wg; t < tms * tns; t += G) gemm_tile(o, t % tms, t / tms, red);
            } break;
            case GOP_CAT:
                op_cat(o, gid, nthreads);
                break;
            case GOP_POOL:
                op_pool(o, gid, nthreads);
                break;
            case GOP_POOL_BWD:
                op_pool_bwd(o, gid, nthreads);
                break;
            case GOP_GATHER_BWD:
                op_gather_bwd(o, gid, nthreads);
                break;
            case GOP_ADD3:
                op_add3(o, gid, nthreads);
                break;
            case GOP_TRANSPOSE: {       // c[k * ldc + n] = a[n * lda + k], n < N, k < K: 32 x 32 tiles through LDS (red as [32][33])
                const int tns = (o.N + 31) >> 5, tks = (o.K + 31) >> 5;
                for (int t = wg; t < tns * tks; t += G) {
                    const int n0 = (t % tns) * 32, k0 = (t / tns) * 32;
                    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;     // 32 x 32 threads
                    __syncthreads();
                    red[y * 33 + x] = (n0 + y < o.N && k0 + x < o.K) ? o.a[(size_t)(n0 + y) * o.lda + k0 + x] : 0.f;
                    __syncthreads();
                    if (k0 + y < o.K && n0 + x < o.N) st_wt(o.c + (size_t)(k0 + y) * o.ldc + n0 + x, red[x * 33 + y]);
                }
            } break;
            default: break;
        }
        if (k + 1 < n_ops) grid_barrier(sync, (unsigned)(k + 1) * (unsigned)G, sync + 1);
    }
}

extern "C" size_t ag2v_graph_op_bytes(void) { return sizeof(GOp); }
// Largest grid whose workgroups are all resident on an otherwise idle device (one 1024-thread workgroup per CU, half the CUs:
// leaves the barrier's polling traffic light and is plenty for <= 180 tiles per operator).
extern "C" int ag2v_graph_program_grid(void) { return 128; }
// Most edges per sample (GOp::T) the POOL / GATHER_BWD operators handle; the program builder refuses larger graphs (they run per operator).
extern "C" int ag2v_graph_program_max_edges(void) { return GP_MAX_EDGES; }
// Runs the program `ops` (device array of n_ops GOp records) with `grid` workgroups (<= ag2v_graph_program_grid()).  sync: two device
// words {arrival counter, timeout flag}; BOTH are zeroed here, so that after the launch a non-zero flag belongs to THIS run (a grid
// barrier gave up -- the workgroups were not co-resident -- and the results are invalid; the caller must read it before trusting them).
extern "C" int ag2v_graph_program_run(const void* ops, int n_ops, unsigned* sync, int grid, void* stream) {
    if (!ops || !sync || n_ops <= 0 || grid <= 0 || grid > 256) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, 2 * sizeof(unsigned), st) != hipSuccess) return AG2V_EHIP;
    hipLaunchKernelGGL(graph_program_kernel, dim3(grid), dim3(GP_THREADS), 0, st, (const GOp*)ops, n_ops, sync);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
