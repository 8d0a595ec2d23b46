// Pointwise activations, loss reductions and the spectral-norm power-iteration step.
//   * LeakyReLU / tanh forward+backward (spade_generator.py:78-80, discriminator.py:368-379)
//   * mean reductions of the GAN losses: hinge (networks/loss.py:57-76), L1 feature matching / flow-warp
//     (loss_model.py:84,97)
//   * torch.nn.utils.spectral_norm train-mode forward: v <- normalize(W^T u), u <- normalize(W v), sigma = u.(W v),
//     W_bar = W / sigma, and its backward  dW = (dW_bar - (sum dW_bar*W_bar) u v^T) / sigma.
#include "common.h"
#include <math.h>

// ---------------------------------------------------------------- activations
// op 0: leaky relu(slope)   op 1: tanh
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int op,
                                                      float slope) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        if (op == 0) {
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        } else {
            v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
        }
        *reinterpret_cast<float4*>(y + i) = v;
    } else {
        for (long long j = i; j < n; ++j) {
            const float t = x[j];
            y[j] = (op == 0) ? (t > 0.f ? t : t * slope) : tanhf(t);
        }
    }
}
// dx = dy * f'(.)  expressed through the forward OUTPUT y: lrelu: y>0 ? 1 : slope ; tanh: 1 - y^2
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                      long long n, int op, float slope) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
        const float4 yy = *reinterpret_cast<const float4*>(y + i), g = *reinterpret_cast<const float4*>(dy + i);
        float4 r;
        if (op == 0) {
            r.x = yy.x > 0.f ? g.x : g.x * slope; r.y = yy.y > 0.f ? g.y : g.y * slope;
            r.z = yy.z > 0.f ? g.z : g.z * slope; r.w = yy.w > 0.f ? g.w : g.w * slope;
        } else {
            r.x = g.x * (1.f - yy.x * yy.x); r.y = g.y * (1.f - yy.y * yy.y);
            r.z = g.z * (1.f - yy.z * yy.z); r.w = g.w * (1.f - yy.w * yy.w);
        }
        *reinterpret_cast<float4*>(dx + i) = r;
    } else {
        for (long long j = i; j < n && j < i + 4; ++j) {
            const float yy = y[j], g = dy[j];
            dx[j] = (op == 0) ? (yy > 0.f ? g : g * slope) : g * (1.f - yy * yy);
        }
    }
}

// ---------------------------------------------------------------- mean reductions
// op 0: sum |a-b|      op 1: sum a            op 2: sum min(a-1,0)      op 3: sum min(-a-1,0)
__device__ __forceinline__ float red_term(int op, float a, float b) {
    switch (op) {
        case 0: return fabsf(a - b);
        case 1: return a;
        case 2: return fminf(a - 1.f, 0.f);
        default: return fminf(-a - 1.f, 0.f);
    }
}
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                             int op, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        s += red_term(op, a[i], b ? b[i] : 0.f);
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ partial, int np, float scale, float* __restrict__ out) {
    __shared__ double shd[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((shd[0] + shd[1] + shd[2] + shd[3]) * (double)scale);
}
// da = gscale * gout[0] * d(term)/da
__global__ __launch_bounds__(256) void reduce_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, int op,
                                                         const float* __restrict__ gout, float scale, float* __restrict__ da) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = gout[0] * scale;
    const float av = a[i];
    float d;
    switch (op) {
        case 0: { const float df = av - b[i]; d = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f); } break;
        case 1: d = 1.f; break;
        case 2: d = (av - 1.f < 0.f) ? 1.f : 0.f; break;
        default: d = (-av - 1.f < 0.f) ? -1.f : 0.f; break;
    }
    da[i] = g * d;
}

// ---------------------------------------------------------------- spectral norm (W viewed as [M, R], row-major)
// v[r] (+)= sum_{m in this row slice} W[m,r] u[m]; grid (ceil(R/256), msplit); msplit > 1 accumulates atomically into zeroed v
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ v,
                                                     int M, int R, int mchunk) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const int m0 = blockIdx.y * mchunk, m1 = min(M, m0 + mchunk);
    float s = 0.f;
    for (int m = m0; m < m1; ++m) s += w[(size_t)m * R + r] * u[m];
    if (gridDim.y > 1) atomicAdd(v + r, s); else v[r] = s;
}
// out[m] = sum_r W[m,r] v[r] : one wave per row
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ out,
                                                    int M, int R) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int r = lane; r < R; r += 64) s += w[(size_t)m * R + r] * v[r];
    s = wave_sum(s);
    if (lane == 0) out[m] = s;
}
// update_u: v[R] holds the un-normalised W^T u and wv[M] = W v of it: v <- v / max(|v|, eps), wv <- wv / max(|v|, eps) (W v is
// linear in v), u <- normalize(wv).  Then sigma = u . wv.   (single workgroup)
// u_used / v_used (nullable): copies of the vectors this call used, for the backward tape (the buffers keep advancing)
__global__ __launch_bounds__(256) void sn_sigma_kernel(float* __restrict__ u, float* __restrict__ wv, float* __restrict__ v, int M,
                                                       int R, float eps, int update_u, float* __restrict__ sigma,
                                                       float* __restrict__ u_used, float* __restrict__ v_used,
                                                       float* __restrict__ inv_sigma = nullptr) {
    __shared__ float sh[4];
    if (update_u) {
        float q = 0.f;
        for (int i = threadIdx.x; i < R; i += 256) q += v[i] * v[i];
        q = block_sum_256(q, sh);
        const float vinv = 1.0f / fmaxf(sqrtf(q), eps);
        for (int i = threadIdx.x; i < R; i += 256) v[i] *= vinv;
        for (int i = threadIdx.x; i < M; i += 256) wv[i] *= vinv;
        __syncthreads();
        float s = 0.f;
        for (int i = threadIdx.x; i < M; i += 256) s += wv[i] * wv[i];
        s = block_sum_256(s, sh);
        const float inv = 1.0f / fmaxf(sqrtf(s), eps);
        for (int i = threadIdx.x; i < M; i += 256) u[i] = wv[i] * inv;
        __syncthreads();
    }
    float d = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) d += u[i] * wv[i];
    d = block_sum_256(d, sh);
    if (threadIdx.x == 0) {
        sigma[0] = d;
        if (inv_sigma) inv_sigma[0] = 1.0f / d;
    }
    if (u_used)
        for (int i = threadIdx.x; i < M; i += 256) u_used[i] = u[i];
    if (v_used)
        for (int i = threadIdx.x; i < R; i += 256) v_used[i] = v[i];
}
__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ wbar,
                                                       long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) wbar[i] = w[i] / sigma[0];
}
// partial[np]: block sums of <dW_bar, W_bar> (dot_partial_kernel); every workgroup adds them up itself (<= 1024 floats from L2,
// in double like reduce_final_kernel) instead of a one-block launch in between
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ dwbar, const float* __restrict__ u, const float* __restrict__ v,
                                                     const float* __restrict__ sigma, const float* __restrict__ partial, int np,
                                                     float* __restrict__ dw, int M, int R, int acc) {
    __shared__ double shd[4];
    double s = 0.0;
    for (int k = threadIdx.x; k < np; k += 256) s += partial[k];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    const float dot = (float)(shd[0] + shd[1] + shd[2] + shd[3]);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * R) return;
    const int m = (int)(i / R), r = (int)(i - (long long)m * R);
    const float g = (dwbar[i] - dot * u[m] * v[r]) / sigma[0];
    dw[i] = acc ? dw[i] + g : g;
}
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                          float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i] * b[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ================================================================ C ABI
extern "C" int ag2v_act_fwd(int op, const float* x, float* y, long long n, float slope, void* stream) {
    if (!x || !y || n <= 0 || op < 0 || op > 1) return AG2V_EINVAL;
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return AG2V_EINVAL;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n, op, slope);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_act_bwd(int op, const float* dy, const float* y, float* dx, long long n, float slope, void* stream) {
    if (!dy || !y || !dx || n <= 0 || op < 0 || op > 1) return AG2V_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, op, slope);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

#define RED_BLOCKS 1024
extern "C" size_t ag2v_reduce_workspace_floats() { return RED_BLOCKS; }

// out[0] = scale * sum_i term_op(a[i], b[i]);   workspace: ag2v_reduce_workspace_floats() floats
extern "C" int ag2v_reduce_fwd(int op, const float* a, const float* b, long long n, float scale, float* out, float* workspace,
                               void* stream) {
    if (!a || !out || !workspace || n <= 0 || op < 0 || op > 3 || (op == 0 && !b)) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int nb = ceil_div(n, 256 * 8);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(nb), dim3(256), 0, st, a, b, n, op, workspace);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, workspace, nb, scale, out);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_reduce_bwd(int op, const float* a, const float* b, long long n, const float* gout, float scale, float* da,
                               void* stream) {
    if (!a || !gout || !da || n <= 0 || op < 0 || op > 3 || (op == 0 && !b)) return AG2V_EINVAL;
    hipLaunchKernelGGL(reduce_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, op, gout, scale, da);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

extern "C" size_t ag2v_spectral_workspace_floats(int M) { return (size_t)M + RED_BLOCKS; }

// W[M,R] (flattened conv weight).  power_iter != 0: one in-place power iteration on u[M], v[R] first (training mode).
// Writes sigma[0] and W_bar = W / sigma.  workspace: ag2v_spectral_workspace_floats(M).
extern "C" int ag2v_spectral_norm_fwd_save(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, float* wbar,
                                           float* sigma, float* u_used, float* v_used, float* workspace, void* stream);
extern "C" int ag2v_spectral_norm_fwd(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, float* wbar,
                                      float* sigma, float* workspace, void* stream) {
    return ag2v_spectral_norm_fwd_save(w, u, v, M, R, eps, power_iter, wbar, sigma, nullptr, nullptr, workspace, stream);
}
// u_used[M], v_used[R] (nullable): the vectors behind this call's sigma, written by the same launches (for the backward tape)
extern "C" int ag2v_spectral_norm_fwd_save(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, float* wbar,
                                           float* sigma, float* u_used, float* v_used, float* workspace, void* stream) {
    if (!w || !u || !v || !wbar || !sigma || !workspace || M <= 0 || R <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* wv = workspace;
    const bool det = (power_iter & 2) != 0;      // AG2V_DETERMINISTIC: W^T u without row splits (no atomics)
    power_iter &= 1;
    if (power_iter) {
        int msplit = 1;
        const int rb = ceil_div(R, 256);
        if (rb < 256 && M >= 64 && !det) { msplit = ceil_div(512, rb); if (msplit > M / 16) msplit = M / 16; if (msplit < 1) msplit = 1; }
        const int mchunk = ceil_div(M, msplit);
        msplit = ceil_div(M, mchunk);
        if (msplit > 1 && hipMemsetAsync(v, 0, (size_t)R * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(rb, msplit), dim3(256), 0, st, w, u, v, M, R, mchunk);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, w, v, wv, M, R);
    hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(256), 0, st, u, wv, v, M, R, eps, power_iter, sigma, u_used, v_used);
    const long long n = (long long)M * R;
    hipLaunchKernelGGL(sn_scale_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, w, sigma, wbar, n);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
// dW = (dW_bar - <dW_bar, W_bar> u v^T) / sigma     (u, v: the vectors the forward used)
// accumulate != 0: dw += (weight_orig is used by every frame pass of a step: one gradient buffer per backward pass)
extern "C" int ag2v_spectral_norm_bwd_acc(const float* dwbar, const float* wbar, const float* u, const float* v, const float* sigma,
                                          int M, int R, float* dw, int accumulate, float* workspace, void* stream) {
    if (!dwbar || !wbar || !u || !v || !sigma || !dw || !workspace || M <= 0 || R <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)M * R;
    int nb = ceil_div(n, 256 * 8);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    float* partial = workspace + M;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, st, dwbar, wbar, n, partial);
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, dwbar, u, v, sigma, partial, nb, dw, M, R,
                       accumulate ? 1 : 0);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
extern "C" int ag2v_spectral_norm_bwd(const float* dwbar, const float* wbar, const float* u, const float* v, const float* sigma,
                                      int M, int R, float* dw, float* workspace, void* stream) {
    return ag2v_spectral_norm_bwd_acc(dwbar, wbar, u, v, sigma, M, R, dw, 0, workspace, stream);
}

// ---------------------------------------------------------------- spectral norm of a frame-batched forward
// The generator runs once per generated frame (generator.py:62-87), i.e. T train-mode forward calls of every spectral-norm
// convolution per step, each advancing (u, v) by one power iteration.  sigma_t depends on W, u, v only -- not on the data -- so
// all T of them are produced up front and the convolution runs ONCE over the [T*B, ...] batch with the raw weight and the
// per-frame-group output scale 1/sigma_t (ConvArgs::gscale).  Same arithmetic per iteration as ag2v_spectral_norm_fwd.
//   sigma[T], inv_sigma[T]; u_used[T*M], v_used[T*R]: the vectors behind sigma_t (backward tape).  power_iter == 0 (eval):
//   T must be 1 (u, v do not move: one sigma).  workspace: ag2v_spectral_workspace_floats(M).
extern "C" int ag2v_spectral_sigmas(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, int T, float* sigma,
                                    float* inv_sigma, float* u_used, float* v_used, float* workspace, void* stream) {
    if (!w || !u || !v || !sigma || !inv_sigma || !workspace || M <= 0 || R <= 0 || T <= 0) return AG2V_EINVAL;
    const bool det = (power_iter & 2) != 0;      // AG2V_DETERMINISTIC: W^T u without row splits (no atomics)
    power_iter &= 1;
    if (!power_iter && T != 1) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* wv = workspace;
    int msplit = 1;
    const int rb = ceil_div(R, 256);
    if (rb < 256 && M >= 64 && !det) { msplit = ceil_div(512, rb); if (msplit > M / 16) msplit = M / 16; if (msplit < 1) msplit = 1; }
    const int mchunk = ceil_div(M, msplit);
    msplit = ceil_div(M, mchunk);
    for (int t = 0; t < T; ++t) {
        if (power_iter) {
            if (msplit > 1 && hipMemsetAsync(v, 0, (size_t)R * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
            hipLaunchKernelGGL(sn_wtu_kernel, dim3(rb, msplit), dim3(256), 0, st, w, u, v, M, R, mchunk);
        }
        hipLaunchKernelGGL(sn_wv_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, w, v, wv, M, R);
        hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(256), 0, st, u, wv, v, M, R, eps, power_iter, sigma + t,
                           u_used ? u_used + (size_t)t * M : nullptr, v_used ? v_used + (size_t)t * R : nullptr, inv_sigma + t);
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// Backward of y_n = scale[g(n)] * conv(x_n, W) with respect to W and scale, from the per-group weight gradients
// G[T][n] = sum_{n in group t} wgrad(dy_n, x_n) (ag2v_conv2d_wgrad_ex, groups = T):
//   dw[i] (+)= sum_t scale[t] * G[t][i];   dscale[t] = <G[t], W>.
// Persistent blocks stride over the weight in chunks of 1024 elements; every thread keeps its share of the (<= 16 per launch) group
// dots in registers and the block reduces them ONCE at the end (one atomic per group and block into the zeroed dscale): a reduction
// per group and chunk made this kernel latency-bound (0.3 TB/s; it reads T x |W| floats).
__global__ __launch_bounds__(256) void sn_group_combine_kernel(const float* __restrict__ G, const float* __restrict__ w,
                                                               const float* __restrict__ scale, int t0, int Tn, long long n,
                                                               float* __restrict__ dw, float* __restrict__ dscale, int acc,
                                                               int vecok) {
    __shared__ float sh[4][16];
    float d[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) d[t] = 0.f;
    float sc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) sc[t] = (t < Tn) ? scale[t0 + t] : 0.f;
    for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (long long)gridDim.x * 1024) {
        const bool vec = vecok && i0 + 3 < n;
        float wv[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            const float4 t4 = *reinterpret_cast<const float4*>(w + i0);
            wv[0] = t4.x; wv[1] = t4.y; wv[2] = t4.z; wv[3] = t4.w;
        } else {
            for (int k = 0; k < 4; ++k) if (i0 + k < n) wv[k] = w[i0 + k];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < Tn) {
                const float* g = G + (size_t)(t0 + t) * n;
                float gv[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec) {
                    const float4 t4 = *reinterpret_cast<const float4*>(g + i0);
                    gv[0] = t4.x; gv[1] = t4.y; gv[2] = t4.z; gv[3] = t4.w;
                } else {
                    for (int k = 0; k < 4; ++k) if (i0 + k < n) gv[k] = g[i0 + k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[k] = fmaf(sc[t], gv[k], o[k]); d[t] = fmaf(gv[k], wv[k], d[t]); }
            }
        }
        if (vec) {
            float4* dst = reinterpret_cast<float4*>(dw + i0);
            if (acc) { const float4 p = *dst; o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
            *dst = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int k = 0; k < 4; ++k) if (i0 + k < n) dw[i0 + k] = acc ? dw[i0 + k] + o[k] : o[k];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float v = wave_sum(d[t]);
        if (lane == 0) sh[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < Tn) atomicAdd(dscale + t0 + threadIdx.x, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
extern "C" int ag2v_sn_group_combine(const float* G, const float* w, const float* scale, int T, long long n, float* dw, int accumulate,
                                     float* dscale, void* stream) {
    if (!G || !w || !scale || !dw || !dscale || T <= 0 || n <= 0) return AG2V_EINVAL;
    // 16-byte accesses need every G[t] row aligned: n % 4 == 0 and aligned bases (else element accesses)
    const int vecok = ((((uintptr_t)G) | ((uintptr_t)w) | ((uintptr_t)dw)) & 15) == 0 && (n & 3) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dscale, 0, (size_t)T * sizeof(float), st) != hipSuccess) return AG2V_EHIP;
    long long nb = ceil_div(n, 1024LL);
    if (nb > 2048) nb = 2048;
    // accumulate is a flag word: AG2V_ACCUMULATE (1) | AG2V_DETERMINISTIC (2: ONE block walks the whole weight, so each dscale[t] is
    // one ordered sum and receives a single add -- slow (one CU), meant for verification runs)
    if (accumulate & 2) nb = 1;
    accumulate &= 1;
    for (int t0 = 0; t0 < T; t0 += 16) {
        const int Tn = (T - t0 < 16) ? T - t0 : 16;
        hipLaunchKernelGGL(sn_group_combine_kernel, dim3((unsigned)nb), dim3(256), 0, st, G, w, scale, t0, Tn, n, dw, dscale,
                           (accumulate || t0 > 0) ? 1 : 0, vecok);
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// Backward of scale[t] = 1 / sigma_t, sigma_t = u_t^T W v_t (u_t, v_t constants):
//   dw[m, r] (+)= sum_t coef_t * u_t[m] * v_t[r],   coef_t = -dscale[t] * scale[t]^2.
__global__ __launch_bounds__(256) void sn_rank_update_kernel(const float* __restrict__ u_used, const float* __restrict__ v_used,
                                                             const float* __restrict__ dscale, const float* __restrict__ scale, int T,
                                                             int M, int R, float* __restrict__ dw, int acc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * R) return;
    const int m = (int)(i / R), r = (int)(i - (long long)m * R);
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
        const float sc = scale[t];
        s = fmaf(-dscale[t] * sc * sc, u_used[(size_t)t * M + m] * v_used[(size_t)t * R + r], s);
    }
    dw[i] = acc ? dw[i] + s : s;
}
extern "C" int ag2v_spectral_sigmas_bwd(const float* dscale, const float* scale, const float* u_used, const float* v_used, int T,
                                        int M, int R, float* dw, int accumulate, void* stream) {
    if (!dscale || !scale || !u_used || !v_used || !dw || T <= 0 || M <= 0 || R <= 0) return AG2V_EINVAL;
    const long long n = (long long)M * R;
    hipLaunchKernelGGL(sn_rank_update_kernel, dim3((unsigned)ceil_div(n, 256LL)), dim3(256), 0, (hipStream_t)stream, u_used, v_used,
                       dscale, scale, T, M, R, dw, accumulate ? 1 : 0);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// ---------------------------------------------------------------- multi-tensor spectral norm (all layers of a network per launch)
// One descriptor per spectral-norm layer; a launch covers every layer of the table, so the T power iterations of a frame-batched
// step cost 4 launches each instead of 4 per layer (38 layers in the generator: 2 280 -> 60 launches per step).
struct SnDesc {
    const float* w;      // [M, R]
    float* u;            // [M]   module buffer weight_u (advanced in place)
    float* v;            // [R]   module buffer weight_v
    float* vtmp;         // [R]   scratch: un-normalised W^T u (zeroed by the caller's memset of the whole scratch block)
    float* wv;           // [M]   scratch: W v
    float* sigma;        // [T]   outputs of this layer
    float* inv_sigma;    // [T]
    float* u_used;       // [T*M] (nullable)
    float* v_used;       // [T*R] (nullable)
    int M, R;
    int blk0;            // first block of this layer in the W^T u launch (blocks = ceil(R/256) * ceil(M/64))
    int row0;            // first row of this layer in the W v launch
};
__device__ __forceinline__ int sn_find(const SnDesc* d, int n, int key, bool by_row) {
    int lo = 0, hi = n - 1;                 // last layer whose first block / row is <= key
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((by_row ? d[mid].row0 : d[mid].blk0) <= key) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// det: the blocks of row chunk 0 walk ALL rows of their 256 columns and store the sum (the other row chunks exit): no atomics
__global__ __launch_bounds__(256) void sn_multi_wtu_kernel(const SnDesc* __restrict__ descs, int n, int det) {
    const int li = sn_find(descs, n, blockIdx.x, false);
    const SnDesc d = descs[li];
    const int local = blockIdx.x - d.blk0;
    const int rb = (d.R + 255) >> 8;
    const int cb = local % rb, mc = local / rb;
    const int r = cb * 256 + threadIdx.x;
    if (r >= d.R) return;
    if (det) {
        if (mc != 0) return;
        float s = 0.f;
        for (int m = 0; m < d.M; ++m) s += d.w[(size_t)m * d.R + r] * d.u[m];
        d.vtmp[r] = s;
        return;
    }
    const int m0 = mc * 64, m1 = min(d.M, m0 + 64);
    float s = 0.f;
    for (int m = m0; m < m1; ++m) s += d.w[(size_t)m * d.R + r] * d.u[m];
    atomicAdd(d.vtmp + r, s);
}
__global__ __launch_bounds__(256) void sn_multi_wv_kernel(const SnDesc* __restrict__ descs, int n, int total_rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int li = sn_find(descs, n, row, true);
    const SnDesc d = descs[li];
    const int m = row - d.row0;
    const int lane = threadIdx.x & 63;
    const float* wr = d.w + (size_t)m * d.R;
    float s = 0.f;
    for (int r = lane; r < d.R; r += 64) s += wr[r] * d.vtmp[r];
    s = wave_sum(s);
    if (lane == 0) d.wv[m] = s;
}
// one workgroup per layer: the normalisations, sigma_t, the buffers' update and the tape copies (as sn_sigma_kernel)
__global__ __launch_bounds__(256) void sn_multi_sigma_kernel(const SnDesc* __restrict__ descs, int t, float eps) {
    __shared__ float sh[4];
    const SnDesc d = descs[blockIdx.x];
    float q = 0.f;
    for (int i = threadIdx.x; i < d.R; i += 256) q += d.vtmp[i] * d.vtmp[i];
    q = block_sum_256(q, sh);
    const float vinv = 1.0f / fmaxf(sqrtf(q), eps);
    for (int i = threadIdx.x; i < d.R; i += 256) {
        const float vn = d.vtmp[i] * vinv;
        d.v[i] = vn;
        if (d.v_used) d.v_used[(size_t)t * d.R + i] = vn;
        d.vtmp[i] = 0.f;                                    // ready for the next iteration's atomics
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < d.M; i += 256) { const float x = d.wv[i] * vinv; s += x * x; }
    s = block_sum_256(s, sh);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    float dot = 0.f;
    for (int i = threadIdx.x; i < d.M; i += 256) {
        const float x = d.wv[i] * vinv;
        const float un = x * inv;
        d.u[i] = un;
        if (d.u_used) d.u_used[(size_t)t * d.M + i] = un;
        dot += un * x;
    }
    dot = block_sum_256(dot, sh);
    if (threadIdx.x == 0) { d.sigma[t] = dot; d.inv_sigma[t] = 1.0f / dot; }
}
extern "C" size_t ag2v_sn_desc_bytes(void) { return sizeof(SnDesc); }
// descs: device array of n_layers SnDesc records (layout above; filled by the caller: blk0 / row0 are the running sums of
// ceil(R/256) * ceil(M/64) and of M), total_blocks / total_rows their totals; the vtmp scratch of every layer must be zero on entry
// (it is left zero on exit).  Runs T training-mode power iterations of every layer.
extern "C" int ag2v_spectral_sigmas_multi(const void* descs, int n_layers, int total_blocks, int total_rows, int T, float eps,
                                          int deterministic, void* stream) {
    if (!descs || n_layers <= 0 || total_blocks <= 0 || total_rows <= 0 || T <= 0) return AG2V_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SnDesc* d = (const SnDesc*)descs;
    for (int t = 0; t < T; ++t) {
        hipLaunchKernelGGL(sn_multi_wtu_kernel, dim3(total_blocks), dim3(256), 0, st, d, n_layers, deterministic ? 1 : 0);
        hipLaunchKernelGGL(sn_multi_wv_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, st, d, n_layers, total_rows);
        hipLaunchKernelGGL(sn_multi_sigma_kernel, dim3(n_layers), dim3(256), 0, st, d, t, eps);
    }
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

// ---------------------------------------------------------------- fused Adam over a flat parameter buffer (K15)
// torch.optim.Adam (scripts/train.py:365-368, meta_models.py:70-72; weight_decay 0, amsgrad off):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float step = lr / bc1;
    if (i + 3 < n) {
        float4 pv = *reinterpret_cast<float4*>(p + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
#define ADAM1(P, G, M, V)                                  \
    M = b1 * M + (1.f - b1) * G;                           \
    V = b2 * V + (1.f - b2) * G * G;                       \
    P -= step * M / (sqrtf(V) / bc2_sqrt + eps);
        ADAM1(pv.x, gv.x, mv.x, vv.x) ADAM1(pv.y, gv.y, mv.y, vv.y) ADAM1(pv.z, gv.z, mv.z, vv.z) ADAM1(pv.w, gv.w, mv.w, vv.w)
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
    } else {
        for (long long j = i; j < n; ++j) {
            float P = p[j], M = m[j], V = v[j];
            const float G = g[j];
            ADAM1(P, G, M, V)
            p[j] = P; m[j] = M; v[j] = V;
        }
    }
#undef ADAM1
}

// One Adam step on flat fp32 buffers (16-byte aligned).  step_index >= 1 (bias corrections computed on the host side of the ABI).
extern "C" int ag2v_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                              float eps, int step_index, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step_index < 1) return AG2V_EINVAL;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return AG2V_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step_index);
    const double bc2 = 1.0 - pow((double)beta2, (double)step_index);
    hipLaunchKernelGGL(adam_kernel, dim3(ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       (float)bc1, (float)sqrt(bc2));
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}
