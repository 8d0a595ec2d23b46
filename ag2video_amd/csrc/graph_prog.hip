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
// producer -> consumer across workgroups, which the barrier orders with an agent-scope release / acquire pair
// (cdna_hip_programming.md, Guideline 16: a workgroup's stores are drained, one lane releases and adds to the arrival counter with an
// agent-scope atomic, polls it, acquires, and the workgroup barrier passes the acquire on to the other waves).
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

#define GP_THREADS 1024
#define GP_WAVES 16
#define GP_SPIN_LIMIT (1u << 24)

// ---------------------------------------------------------------- grid barrier (agent-scope release / acquire)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, unsigned* timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave: its stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > GP_SPIN_LIMIT) { __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// ---------------------------------------------------------------- GEMM
// one 32 x 32 tile: wave w takes k in [w * Kw, (w + 1) * Kw), Kw a multiple of 8; lane (r = lane & 31, h = lane >> 5) loads
// A[m0 + r][k8 + 4 h .. + 3] and B[n0 + r][k8 + 4 h .. + 3] for every k8 of its slice at once; MFMA j of a k8 step multiplies the
// k = k8 + j (h = 0) and k = k8 + 4 + j (h = 1) columns.  Partials meet in LDS [16][1024].
#define GP_MAXSTEP 8        // k8 steps per wave held in registers at once (K <= 16 * 8 * 8 = 1024 in one batch; longer K loops)
__device__ __forceinline__ void gemm_tile(const GOp& o, int tm, int tn, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m = tm * 32 + r, n = tn * 32 + r;
    const int K = o.K;
    const int Kw = ((K + GP_WAVES * 8 - 1) / (GP_WAVES * 8)) * 8;
    const int k_lo = wave * Kw, k_hi = min(K, k_lo + Kw);
    const bool mok = m < o.M, nok = n < o.N;
    const float* ap = o.a + (size_t)(mok ? m : 0) * o.lda;
    const float* mp = o.mask ? o.mask + (size_t)(mok ? m : 0) * o.ldm : nullptr;
    const float* bp = o.b + (size_t)(nok ? n : 0) * o.ldb;
    const bool vec = ((o.lda | o.ldb | (o.mask ? o.ldm : 0)) & 3) == 0 &&
                     ((((uintptr_t)o.a) | ((uintptr_t)o.b) | ((uintptr_t)(o.mask ? o.mask : o.a))) & 15) == 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int kb = k_lo; kb < k_hi; kb += 8 * GP_MAXSTEP) {
        float4 av[GP_MAXSTEP], bv[GP_MAXSTEP];
#pragma unroll
        for (int s = 0; s < GP_MAXSTEP; ++s) {
            const int k = kb + 8 * s + 4 * h;
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
            if (k < k_hi) {
                if (vec && k + 3 < k_hi) {
                    if (mok) {
                        a4 = *reinterpret_cast<const float4*>(ap + k);
                        if (mp) {
                            const float4 q = *reinterpret_cast<const float4*>(mp + k);
                            a4.x = q.x > 0.f ? a4.x : 0.f; a4.y = q.y > 0.f ? a4.y : 0.f;
                            a4.z = q.z > 0.f ? a4.z : 0.f; a4.w = q.w > 0.f ? a4.w : 0.f;
                        }
                    }
                    if (nok) b4 = *reinterpret_cast<const float4*>(bp + k);
                } else {
                    float at[4] = {0.f, 0.f, 0.f, 0.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int j = 0; j < 4; ++j)
                        if (k + j < k_hi) {
                            if (mok) at[j] = (mp && !(mp[k + j] > 0.f)) ? 0.f : ap[k + j];
                            if (nok) bt[j] = bp[k + j];
                        }
                    a4 = make_float4(at[0], at[1], at[2], at[3]);
                    b4 = make_float4(bt[0], bt[1], bt[2], bt[3]);
                }
            }
            av[s] = a4; bv[s] = b4;
        }
#pragma unroll
        for (int s = 0; s < GP_MAXSTEP; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].x, bv[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].y, bv[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].z, bv[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s].w, bv[s].w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                                   // the previous tile's reduction reads are done
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
        *dst = o.acc ? *dst + s : s;
    }
}

// ---------------------------------------------------------------- row operators: one wave per row, lanes along the columns
__device__ __forceinline__ void op_cat(const GOp& o, int row) {
    const int lane = threadIdx.x & 63;
    float* dst = o.c + (size_t)row * o.ldc;
    const float* s0 = o.a + (size_t)(o.i0 ? o.i0[row] : row) * o.lda;
    for (int i = lane; i < o.w0; i += 64) dst[i] = s0[i];
    if (o.b) {
        const float* s1 = o.b + (size_t)(o.i1 ? o.i1[row] : row) * o.ldb;
        for (int i = lane; i < o.w1; i += 64) dst[o.w0 + i] = s1[i];
    }
    if (o.d) {
        const float* s2 = o.d + (size_t)(o.i2 ? o.i2[row] : row) * o.ldd;
        for (int i = lane; i < o.w2; i += 64) dst[o.w0 + o.w1 + i] = s2[i];
    }
}
// pooled[bo, :Hd] = (sum over valid edges t of sample b with s(t) == bo of newt[t, :Hd] + ... o(t) == bo of newt[t, Hd + Dp:]) / count
__device__ __forceinline__ void op_pool(const GOp& o, int bo) {
    const int lane = threadIdx.x & 63;
    const int b = bo / o.O, Hd = o.w0, Dp = o.w1;
    int cnt = 0;
    for (int t = 0; t < o.T; ++t) {
        const int e = b * o.T + t;
        if (!o.valid[e]) continue;
        cnt += (o.i0[e] == bo) + (o.i1[e] == bo);
    }
    for (int i = lane; i < Hd; i += 64) {
        float acc = 0.f;
        for (int t = 0; t < o.T; ++t) {
            const int e = b * o.T + t;
            if (!o.valid[e]) continue;
            if (o.i0[e] == bo) acc += o.a[(size_t)e * o.lda + i];
            if (o.i1[e] == bo) acc += o.a[(size_t)e * o.lda + Hd + Dp + i];
        }
        o.c[(size_t)bo * o.ldc + i] = cnt > 0 ? acc / (float)cnt : acc;
    }
    if (lane == 0) const_cast<float*>(o.mask)[bo] = (float)cnt;          // counts (written here, read by POOL_BWD of the backward program)
}
__device__ __forceinline__ void op_pool_bwd(const GOp& o, int bt) {
    const int lane = threadIdx.x & 63;
    const int Hd = o.w0, Dp = o.w1;
    const int s = o.i0[bt], ob = o.i1[bt];
    const bool v = o.valid[bt] != 0;
    const float cs = o.mask[s], co = o.mask[ob];
    const float is = (v && cs > 0.f) ? 1.0f / cs : 0.f, io = (v && co > 0.f) ? 1.0f / co : 0.f;
    const float* ds = o.a + (size_t)s * o.lda;
    const float* dob = o.a + (size_t)ob * o.lda;
    float* dst = o.c + (size_t)bt * o.ldc;
    for (int i = lane; i < 2 * Hd + Dp; i += 64) {
        float g;
        if (i < Hd) g = v ? ds[i] * is : 0.f;
        else if (i < Hd + Dp) g = o.b ? o.b[(size_t)bt * o.ldb + (i - Hd)] : 0.f;
        else g = v ? dob[i - Hd - Dp] * io : 0.f;
        dst[i] = g;
    }
}
// dobj[bo, :Din] (+)= res[bo] + sum_t (s(t) == bo) dtri[t, :Din] + (o(t) == bo) dtri[t, Din + Dp:]   (every edge: the gather is unmasked)
__device__ __forceinline__ void op_gather_bwd(const GOp& o, int bo) {
    const int lane = threadIdx.x & 63;
    const int b = bo / o.O, Din = o.w0, Dp = o.w1;
    for (int i = lane; i < Din; i += 64) {
        float acc = o.res ? o.res[(size_t)bo * o.ldr + i] : 0.f;
        for (int t = 0; t < o.T; ++t) {
            const int e = b * o.T + t;
            if (o.i0[e] == bo) acc += o.a[(size_t)e * o.lda + i];
            if (o.i1[e] == bo) acc += o.a[(size_t)e * o.lda + Din + Dp + i];
        }
        o.c[(size_t)bo * o.ldc + i] = acc;
    }
}
__device__ __forceinline__ void op_add3(const GOp& o, int row) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < o.N; i += 64) {
        float v = o.a[(size_t)row * o.lda + i];
        if (o.b) v += o.b[(size_t)row * o.ldb + i];
        if (o.d) v += o.d[(size_t)row * o.ldd + i];
        o.c[(size_t)row * o.ldc + i] = v;
    }
}

__global__ __launch_bounds__(GP_THREADS) void graph_program_kernel(const GOp* __restrict__ ops, int n_ops, unsigned* __restrict__ sync) {
    __shared__ float red[GP_WAVES * 1024];
    __shared__ GOp cur;
    const int G = gridDim.x, wg = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    for (int k = 0; k < n_ops; ++k) {
        // the descriptor through LDS: one copy per workgroup, then uniform reads (the array is constant for the launch)
        __syncthreads();
        if (threadIdx.x < (int)(sizeof(GOp) / 4)) reinterpret_cast<unsigned*>(&cur)[threadIdx.x] = reinterpret_cast<const unsigned*>(ops + k)[threadIdx.x];
        __syncthreads();
        const GOp& o = cur;
        switch (o.type) {
            case GOP_GEMM: {
                const int tms = (o.M + 31) >> 5, tns = (o.N + 31) >> 5;
                for (int t = wg; t < tms * tns; t += G) gemm_tile(o, t % tms, t / tms, red);
            } break;
            case GOP_CAT:
                for (int r = wg * GP_WAVES + wave; r < o.M; r += G * GP_WAVES) op_cat(o, r);
                break;
            case GOP_POOL:
                for (int r = wg * GP_WAVES + wave; r < o.M; r += G * GP_WAVES) op_pool(o, r);
                break;
            case GOP_POOL_BWD:
                for (int r = wg * GP_WAVES + wave; r < o.M; r += G * GP_WAVES) op_pool_bwd(o, r);
                break;
            case GOP_GATHER_BWD:
                for (int r = wg * GP_WAVES + wave; r < o.M; r += G * GP_WAVES) op_gather_bwd(o, r);
                break;
            case GOP_ADD3:
                for (int r = wg * GP_WAVES + wave; r < o.M; r += G * GP_WAVES) op_add3(o, r);
                break;
            case GOP_TRANSPOSE: {       // c[k * ldc + n] = a[n * lda + k], n < N, k < K: 32 x 32 tiles through LDS (red as [32][33])
                const int tns = (o.N + 31) >> 5, tks = (o.K + 31) >> 5;
                for (int t = wg; t < tns * tks; t += G) {
                    const int n0 = (t % tns) * 32, k0 = (t / tns) * 32;
                    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;     // 32 x 32 threads
                    __syncthreads();
                    red[y * 33 + x] = (n0 + y < o.N && k0 + x < o.K) ? o.a[(size_t)(n0 + y) * o.lda + k0 + x] : 0.f;
                    __syncthreads();
                    if (k0 + y < o.K && n0 + x < o.N) o.c[(size_t)(k0 + y) * o.ldc + n0 + x] = red[x * 33 + y];
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
// Runs the program `ops` (device array of n_ops GOp records) with `grid` workgroups (<= ag2v_graph_program_grid()).  sync: two device
// words {arrival counter, timeout flag}; the counter is zeroed here, the flag is left for the caller to read (non-zero: a grid barrier
// gave up -- the workgroups were not co-resident -- and the results are invalid).
extern "C" int ag2v_graph_program_run(const void* ops, int n_ops, unsigned* sync, int grid, void* stream) {
    if (!ops || !sync || n_ops <= 0 || grid <= 0 || grid > 256) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, sizeof(unsigned), st) != hipSuccess) return AG2V_EHIP;
    hipLaunchKernelGGL(graph_program_kernel, dim3(grid), dim3(GP_THREADS), 0, st, (const GOp*)ops, n_ops, sync);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
