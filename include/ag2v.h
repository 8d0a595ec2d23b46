/* ag2v.h -- C ABI of libag2v.so: the MI355X (gfx950) kernels of the AG2Video training step.
 *
 * The reference (roeiherz/AG2Video) has no FFI for this path: it is plain torch.nn modules whose arithmetic is
 * stock ATen.  Each entry point below replaces the ATen call site(s) cited next to it; file:line are
 * relative to the reference checkout.  Conventions:
 *   - every tensor is a raw device pointer to contiguous fp32 (NCHW for images) unless stated (int64 indices, uint8/int32 masks);
 *   - the caller owns all memory; no entry point allocates, frees or synchronises; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*) and are safe to capture in a hipGraph;
 *   - the library holds no process-global mutable state: every option travels with the call (ag2v_conv_opts, ag2v_halo_opts); a call
 *     that leaves precision / min_rows unnamed takes the defaults of the DEVICE it runs on (a per-device record initialised from the
 *     environment).  The ag2v_set_* functions are deprecated shims that edit that record for the calling thread's current device;
 *   - scratch memory is passed in as `workspace` and sized by the matching ag2v_*_workspace_floats();
 *   - return value: 0 on success, AG2V_EINVAL (-1) for a rejected argument, AG2V_EHIP (-2) for a HIP launch error
 *     (the Python host turns non-zero into RuntimeError, matching the reference's "raise, never abort" convention,
 *     scripts/train.py:466-468).
 */
#ifndef AG2V_H
#define AG2V_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AG2V_OK 0
#define AG2V_EINVAL (-1)
#define AG2V_EHIP (-2)

/* ---- convolution: nn.Conv2d forward/backward (aten.convolution) ------------------------------------------------
 * call sites: generator.py:30; flows_generator.py:32-54,82-85; spade_generator.py:16-34; architecture.py:29-32;
 * normalization.py:90-94; discriminator.py:368-379.   x[N,C,H,W], w[K,C,R,S], y[N,K,OH,OW], OH=(H+2p-R)/s+1.
 * fwd: y = conv(x,w) (+bias[K]) (+residual, same shape as y).  dgrad: dx from dy.  wgrad: dw from dy,x. */
/* Per-call options of the *_ex convolution entry points (plain C struct, passed by pointer; NULL = all defaults).
 *   precision: arithmetic of the 128/256-row MFMA tiles: 1 = bf16x3 split products with fp32 accumulation (each fp32 operand
 *              = bf16 hi + bf16 lo, a*b = hi*hi + hi*lo + lo*hi: ~1e-5 relative per product), 0 = exact fp32 MFMA,
 *              2 = (forward only; backward entry points treat it as 1) the same three products on FP16 hi / lo parts
 *              (11 + 11 significand bits: ~5e-6 of rms(out), the level of a sequential fp32 fmaf chain) for operands of O(1)
 *              magnitude -- below ~1e-2 the lo part is an fp16 subnormal and the accuracy decays, so the weights are packed
 *              times 32 (exact, undone in the epilogue) and the caller uses it on normalised activations only (the flow network),
 *              < 0 = the current device's default (AG2V_CONV_PRECISION, initially 1; deprecated ag2v_set_conv_precision);
 *   min_rows:  in bf16x3 mode, GEMMs with fewer rows (output channels; input channels for the data gradient) stay on the exact
 *              fp32 tiles; <= 0 = the current device's default (AG2V_BF16X3_MIN_M, initially 128; deprecated ag2v_set_bf16x3_min_rows);
 *   gscale:    nullable device pointer: the convolution product of sample n is multiplied by gscale[n / group_n] before bias
 *              and residual are added (forward and data gradient).  This is how a spectral-norm convolution of the
 *              frame-batched generator runs: torch.nn.utils.spectral_norm divides W by sigma_t in the t-th forward call of a
 *              step (call sites architecture.py:35-39, flows_generator.py:79-85, normalization.py:26-27 inside the frame loop
 *              generator.py:62-87); with all T frames in one [T*B, ...] batch the kernel convolves the raw weight and scales
 *              group t by 1/sigma_t;
 *   kernel:    OUT: name (as rocprofv3 prints it) of the main kernel the call launched (static string);
 *   in_split:  (forward: x; weight gradient: x) the tensor holds SPLIT WORDS -- (bf16(v) << 16) | bf16(v - bf16(v)) per value, as
 *              ag2v_norm_apply_g(out_split = 1) writes them -- instead of fp32: the bf16x3 kernels then skip their own split
 *              (bit-identical results).  Only where ag2v_conv2d_split_input_ok() says so; otherwise AG2V_EINVAL;
 *   dy_split:  (data gradient, weight gradient) the same for the output gradient dy (written split by ag2v_norm_bwd_gs /
 *              ag2v_spade_pair_bwd_s); only where ag2v_conv2d_split_dy_ok() says so;
 *   out_split: (forward, bf16x3 path, no residual) the result is stored as split words by the epilogue -- for an activation that
 *              only convolutions taking in_split read (SPADE: relu(mlp_shared(seg)) feeds the gamma | beta convolution);
 *   deterministic: != 0: bit-reproducible results -- NO floating-point atomics.  Forward / data gradient: the reduction of an output
 *              tile is never split over workgroups (small maps run on fewer CUs).  Weight gradient: every pixel-range split of a
 *              launch stores its partial gradient into det_workspace and an ordered pass adds the partials (and, when accumulating,
 *              the previous dw) in split order.  The default (0) accumulates split partials with fp32 atomics, whose order varies
 *              from run to run (~1e-7 relative per sum);
 *   det_workspace / det_workspace_floats: (weight gradient) caller scratch of at least ag2v_conv2d_wgrad_det_workspace_floats(...)
 *              floats (may be NULL when that returns 0): required in deterministic mode; in the default mode the LDS-DMA kernel
 *              (3x3 stride-1 on split words) uses it for the same slab reduction when given -- results are then bit-reproducible for
 *              that launch too -- and falls back to atomics without it;
 *   out_nstride: (data gradient, stride 1, not accumulating, more than 4 input channels) floats between consecutive samples of dx;
 *              0 = C * H * W, the dense tensor.  Larger: dx is the channel slice [:, c0 : c0 + C] of a wider [N, Ctot, H, W] buffer
 *              (out_nstride = Ctot * H * W, dx = buffer + c0 * H * W) -- the gradients of several convolutions that share a
 *              consumer are written straight into the concatenation that consumer's merged weight / data gradient reads
 *              (conv2d_shared_input: every SPADE's mlp_shared, normalization.py:90-94).  The mask keeps the dense layout.
 *              Anything else with a non-dense stride: AG2V_EINVAL;
 *   out_mstride: (forward only; 0 = off) SEVERAL convolutions of one input as ONE launch: w / bias hold K = n * 128 output channels (the
 *              n weight sets concatenated), and every 128 of them are written as their own dense [N, 128, OH, OW] tensor, out_mstride
 *              floats apart (y + i * out_mstride; out_mstride >= N * 128 * OH * OW).  The SPADE layers of a residual block read the same
 *              resized segmentation map through their own 128-channel mlp_shared convolution (normalization.py:90-94,
 *              architecture.py:52-62): the gathered map is then staged once for all of them.  Only where ag2v_conv2d_fwd_mtile_ok()
 *              says so (the LDS-DMA kernels' 128-row tile); otherwise AG2V_EINVAL, before anything is launched;
 *   out_chan_sums / chan_sums_done: (data gradient) out_chan_sums, when not NULL, points at C floats the caller has initialised; a launch
 *              that runs on the LDS-DMA kernels (3x3 stride-1, not deterministic) then ADDS the per-channel sums of the dx it writes
 *              (over samples and pixels; of the values, also when dx is written as split words) from its epilogue -- the bias gradient of
 *              the convolution that produced the data-gradient's input (reference: autograd's sum over dy for nn.Conv2d's bias) without
 *              another pass over dx -- and sets chan_sums_done = 1.  Any other launch leaves the buffer untouched and sets
 *              chan_sums_done = 0: the caller then sums dx itself (ag2v_channel_sum_acc_s).  Float atomics: ~1e-7 relative run to run. */
typedef struct ag2v_conv_opts {
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
} ag2v_conv_opts;
/* Defaults of calls that do not name precision / min_rows: a record per DEVICE, initialised from AG2V_CONV_PRECISION /
 * AG2V_BF16X3_MIN_M.  The setters are deprecated shims editing the calling thread's current device's record (the product passes both
 * fields with every call and never reads it). */
int ag2v_set_conv_precision(int mode);
int ag2v_get_conv_precision(void);
int ag2v_set_bf16x3_min_rows(int rows);
int ag2v_get_bf16x3_min_rows(void);
/* Diagnostic switches of the halo-tile launchers (AG2V_NO_HALO, AG2V_HALO_MFMA32, AG2V_HALO_MIN_TILES, AG2V_NO_HALO_WIDE, AG2V_NO_WGRAD_HALO: which kernel
 * FAMILY takes a launch -- same products, another summation order; each has a test on its non-default side) are read
 * from the environment once at load time; ag2v_halo_reload_env() reads them again (tests / tools, with no launch in flight).
 * ag2v_halo_min_tiles(): the tile count from which a 3x3 stride-1 launch takes those kernels (384 unless AG2V_HALO_MIN_TILES). */
int ag2v_halo_reload_env(void);
int ag2v_halo_min_tiles(void);
/* Name of the main kernel launched by the most recent ag2v_conv2d_* call of the calling thread (ag2v_conv_opts::kernel is the
 * per-call form). */
const char* ag2v_last_conv_kernel(void);
size_t ag2v_conv2d_workspace_floats(int C, int K, int R, int S);  /* k-major copy of the (sub-)filters; fwd and dgrad */
int ag2v_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                    int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace, void* stream);
int ag2v_conv2d_dgrad(const float* dy, const float* w, float* dx,
                      int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace, void* stream);
/* ag2v_conv2d_fwd with y = max(., 0) when relu != 0 (SPADE's mlp_shared = Conv2d + ReLU, normalization.py:92-95), and
 * ag2v_conv2d_dgrad with dx = (mask > 0) ? dx : 0 (mask nullable, shaped like dx): the backward of the ReLU that produced this
 * convolution's input, applied in the epilogue (or in one pass over dx when the launch splits K / the stride has several classes). */
int ag2v_conv2d_fwd_act(const float* x, const float* w, const float* bias, const float* residual, float* y,
                        int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, float* workspace, void* stream);
int ag2v_conv2d_dgrad_masked(const float* dy, const float* w, const float* mask, float* dx,
                             int N, int C, int H, int W, int K, int R, int S, int stride, int pad, float* workspace, void* stream);
/* accumulate != 0: dx += instead of dx = (several convolutions reading one tensor -- the SPADE mlp_shared convolutions of a residual
 * block, normalization.py:92-95 applied three times to the same segmap -- sum their input gradients in place).  mask must be null then. */
int ag2v_conv2d_dgrad_acc(const float* dy, const float* w, const float* mask, float* dx,
                          int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int accumulate, float* workspace,
                          void* stream);
int ag2v_conv2d_wgrad(const float* dy, const float* x, float* dw,
                      int N, int C, int H, int W, int K, int R, int S, int stride, int pad, void* stream);
/* The full forms (what ag2video_amd/ops.py calls).  Batches whose gathered operand exceeds 4 GiB are processed in sample chunks
 * inside the call.
 *   fwd_ex:   y = conv(x, w) * gscale[n / group_n] + bias + residual, relu != 0: y = max(y, 0).
 *   dgrad_ex: dx (+)= dgrad(dy, w) * gscale[n / group_n], masked as ag2v_conv2d_dgrad_masked; workspace_floats =
 *             ag2v_conv2d_dgrad_workspace_floats(...) (the weight copy plus, for <= 4 input channels, the [N, C R S, OH, OW]
 *             intermediate of the 1x1-GEMM + shifted-plane-sum form); with less workspace the general kernels run.
 *   wgrad_ex: groups == 1: dw (+)= wgrad(dy, x).  groups > 1 (N % groups == 0): dw is [groups][K, C, R, S] and receives one
 *             weight gradient per group of N / groups consecutive samples -- dL/dWbar_t of every frame t, which the backward of
 *             spectral norm needs separately (ag2v_sn_group_combine). */
int ag2v_conv2d_fwd_ex(const float* x, const float* w, const float* bias, const float* residual, float* y,
                       int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, float* workspace,
                       void* stream, ag2v_conv_opts* opts);
size_t ag2v_conv2d_dgrad_workspace_floats(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
int ag2v_conv2d_dgrad_ex(const float* dy, const float* w, const float* mask, float* dx,
                         int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int accumulate,
                         float* workspace, size_t workspace_floats, void* stream, ag2v_conv_opts* opts);
int ag2v_conv2d_wgrad_ex(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S,
                         int stride, int pad, int accumulate, int groups, void* stream, ag2v_conv_opts* opts);
size_t ag2v_conv2d_wgrad_det_workspace_floats(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int groups,
                                              const ag2v_conv_opts* opts);
int ag2v_conv2d_split_input_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows_fwd,
                               int min_rows_bwd, int precision);
/* 1 when ag2v_conv2d_fwd_ex accepts ag2v_conv_opts::out_mstride for this call (K % 128 == 0, 3x3 stride-1, every sample chunk on the
 * LDS-DMA kernels' 128-row tile), else 0.  Reference call site: the mlp_shared convolutions of a block, normalization.py:90-94. */
int ag2v_conv2d_fwd_mtile_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows, int precision,
                             int in_split, int out_split, int deterministic);
int ag2v_conv2d_split_dy_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int min_rows_bwd, int precision);
/* accumulate != 0: dw += instead of dw = (a weight used by several forward calls of one step collects its gradient in one buffer). */
int ag2v_conv2d_wgrad_acc(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R, int S,
                          int stride, int pad, int accumulate, void* stream);
/* Flag word of the `accumulate` (and, for the spectral-norm entry points, `power_iter`) arguments of the reductions below:
 * bit 0 = the argument's own meaning (accumulate / run the power iteration), bit 1 = AG2V_DETERMINISTIC: ordered reduction without
 * floating-point atomics (one workgroup per channel / no row splits / one block) -- slower, bit-reproducible. */
#define AG2V_ACCUMULATE 1
#define AG2V_DETERMINISTIC 2
/* out[c] = sum over n,hw of x[n,c,hw]: conv bias gradient */
int ag2v_channel_sum(const float* x, float* out, int N, int C, int HW, void* stream);
/* accumulate & 1: out[c] += ... (one gradient buffer per bias and backward pass; the generator runs once per frame); & 2: deterministic */
int ag2v_channel_sum_acc(const float* x, float* out, int N, int C, int HW, int accumulate, void* stream);
/* ... over a tensor of split words when x_split != 0 (each value is hi + lo). */
int ag2v_channel_sum_acc_s(const float* x, float* out, int N, int C, int HW, int accumulate, int x_split, void* stream);

/* ---- batch norm / SPADE / instance norm -------------------------------------------------------------------------
 * F.batch_norm(training) in sync_batchnorm/batchnorm.py:63-68; SPADE.forward normalization.py:96-110;
 * nn.InstanceNorm2d + LeakyReLU discriminator.py:374 (via normalization.py:43-44). */
size_t ag2v_bn_workspace_floats(int N, int C, int HW);
int ag2v_bn_stats(const float* x, int N, int C, int HW, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, float* workspace, void* stream);
/* mode 0: y = lrelu((x-mean)*invstd*w[c] + b[c], slope); mode 1: y = lrelu(xhat*(1+gamma)+beta, slope); slope 1 = identity.
 * mode 2 = mode 1 with gamma and beta (and, in ag2v_norm_bwd, dgamma and dbeta) being the two channel halves of one
 * [N, 2C, HW] tensor: pass the pointers of the halves' first elements; element (n, c, i) is read at n*2C*HW + c*HW + i. */
int ag2v_norm_apply(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                    const float* gamma, const float* beta, float* y, int N, int C, int HW, float slope, void* stream);
int ag2v_norm_bwd(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                  const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                  int N, int C, int HW, float slope, float* workspace, void* stream);
/* Sample groups: the T frame passes of the reference's generator loop (generator.py:62-87) are T separate F.batch_norm calls
 * over B samples each; the frame-batched form holds them as one [T*B, C, HW] tensor, Ng = B samples per group.  mean / invstd
 * are [N / Ng, C] (sample n uses row n / Ng); the running statistics receive the N / Ng momentum updates in group order; sums is
 * [N / Ng, C, 4] = (sum dxhat, sum dxhat xhat, sum dbeta, sum dgamma): mode 0 -> the first two are (dbias, dweight) of the affine
 * parameters; modes 1 / 2 -> the last two are the channel sums of the gamma / beta gradients (the bias gradient of the convolution
 * that produced them).  Ng == N is the plain form above. */
int ag2v_bn_stats_g(const float* x, int N, int C, int HW, int Ng, float eps, float momentum, float* mean, float* invstd,
                    float* running_mean, float* running_var, float* workspace, void* stream);
/* The second half of ag2v_bn_stats_g on partials written by the kernel that produced x: partial[G][C][nparts][3] = (count, mean, M2) of
 * disjoint element sets covering each (group, channel) -- sync_batchnorm/batchnorm.py:63-68 F.batch_norm's statistics, combined in double. */
int ag2v_bn_finalize_partials(const float* partial, int G, int C, int nparts, float eps, float momentum, float* mean, float* invstd,
                              float* running_mean /*nullable*/, float* running_var /*nullable*/, void* stream);
int ag2v_norm_apply_g(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                      const float* gamma, const float* beta, float* y, int N, int C, int HW, int Ng, float slope, void* stream);
/* ag2v_norm_apply_g whose output is stored as split words when out_split != 0 (see ag2v_conv_opts::in_split): for a tensor that
 * only bf16x3 convolutions read (generator.py: the normalised layout features every SPADE block convolves). */
int ag2v_norm_apply_gs(int mode, const float* x, const float* mean, const float* invstd, const float* w, const float* b,
                       const float* gamma, const float* beta, float* y, int N, int C, int HW, int Ng, float slope, int out_split,
                       void* stream);
/* ag2v_norm_apply_gs(mode 0, ...) that also writes y_half [N, C, H/2, W/2] = y[:, :, ::2, ::2] (same word format) from the registers:
 * the first level of the decoder's segmentation-map pyramid (normalization.py:102, F.interpolate(segmap, mode='nearest')) without a
 * second pass over y.  H % 2 == 0, W % 8 == 0, 16-byte aligned tensors. */
int ag2v_norm_apply_half(const float* x, const float* mean, const float* invstd, const float* w, const float* b, float* y,
                         float* y_half, int N, int C, int H, int W, int Ng, float slope, int out_split, void* stream);
int ag2v_norm_bwd_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                    const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                    int N, int C, int HW, int Ng, float slope, float* workspace, void* stream);
/* ag2v_norm_bwd_g whose dgamma / dbeta (modes 1 / 2) are stored as split words when dgb_split != 0 (ag2v_conv_opts::dy_split): the
 * gradient of the convolution that produced gamma | beta goes into that convolution's data gradient and weight gradient only. */
int ag2v_norm_bwd_gs(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                     const float* bias, const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                     int N, int C, int HW, int Ng, float slope, float* workspace, int dgb_split, void* stream);
/* ... in phases, for data-parallel SYNCHRONISED batch norm (sync_batchnorm/batchnorm.py:105-126, _compute_mean_std :128-145) on the fused
 * paths: phase 0 = both passes (the function above), 1 = the reduce pass only (sums written), 2 = the apply pass only (sums read).
 * The caller runs 1, all-reduces columns 0 / 1 of sums[N / Ng, C, 4] (sum dxhat, sum dxhat xhat) over the ranks -- columns 2 / 3,
 * the channel sums of dbeta / dgamma, stay per rank -- and runs 2 with inv_count = 1 / (elements per (group, channel) over ALL
 * ranks); inv_count <= 0 = this process's own count. */
int ag2v_norm_bwd_gs_p(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                       const float* bias, const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta, float* sums,
                       int N, int C, int HW, int Ng, float slope, float* workspace, int dgb_split, int phase, float inv_count,
                       void* stream);
/* (mode 0, y == NULL: the LeakyReLU branch is recomputed from x, w and `bias` -- the affine pair of the forward pass -- instead of
 *  being read from the forward output: 5 instead of 7 tensor passes.)
 * (mode 3 = mode 2 whose `gamma` points at a COMPACT copy [N, C, HW] of the gamma half; dgamma / dbeta are still the halves of one
 *  [N, 2C, HW] tensor.  The backward never reads beta, so a caller that keeps this copy instead of the gamma | beta tensor halves
 *  that part of its tape: ops.set_lean_tape.) */
/* ag2v_norm_bwd_gs(mode 0, y == NULL) for an output that was read directly and through nearest-neighbour down-samplings (the
 * decoder's segmentation-map pyramid, normalization.py:102): dy = g_full (nullable) + sum_l scatter(g_levels[l]) is formed while
 * loading.  g_levels / hs / ws: host arrays of n_levels <= 8 entries (null level pointers are skipped). */
int ag2v_norm_bwd_pyr(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                      const float* x, const float* w, const float* bias, const float* mean, const float* invstd, float* dx,
                      float* sums, int N, int C, int H, int W, int Ng, float slope, float* workspace, void* stream);
int ag2v_norm_bwd_pyr_p(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                        const float* x, const float* w, const float* bias, const float* mean, const float* invstd, float* dx,
                        float* sums, int N, int C, int H, int W, int Ng, float slope, float* workspace, int phase, float inv_count,
                        void* stream);      /* phases as ag2v_norm_bwd_gs_p */
/* ag2v_norm_bwd_g in two steps, for data-parallel SYNCHRONISED batch norm (sync_batchnorm/batchnorm.py:105-126): the caller
 * all-reduces sums[N / Ng, C, 4] over the ranks between them and passes inv_count = 1 / (elements per (group, channel) over ALL ranks). */
int ag2v_norm_bwd_reduce_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* mean,
                           const float* invstd, float* sums, int N, int C, int HW, int Ng, float slope, float* workspace, void* stream);
int ag2v_norm_bwd_apply_g(int mode, const float* dy, const float* y, const float* x, const float* gamma, const float* w,
                          const float* mean, const float* invstd, const float* sums, float inv_count, float* dx, float* dgamma,
                          float* dbeta, int N, int C, int HW, int Ng, float slope, void* stream);
/* SPADE pair: the two modulations a SPADEResnetBlock with a learned shortcut applies to ONE input (norm_0 and norm_s,
 * architecture.py:52-53,60-61) in one pass, the input read at its resolution BEFORE nn.Upsample(scale_factor=2)
 * (spade_generator.py:62-76; up = 2: element (y, x) is xs[y/2][x/2]; up = 1: no upsampling).  xs [N, C, H/up, W/up]; gb0 / gb1
 * [N, 2C, H, W] (gamma half, beta half); y0 / y1 [N, C, H, W]; mean / invstd [N / Ng, C]; sums [N / Ng, C, 8]; dxs like xs (both
 * branches' input gradients summed and reduced over the 2 x 2 replica blocks).  _stats: rep = up * up; both running pairs (nullable)
 * receive the per-group updates.  ag2v_spade_pair_supported(H, W, up) != 0 is required (W % 4 == 0; up = 2: 4096 % (2 W) == 0). */
int ag2v_spade_pair_supported(int H, int W, int up);
size_t ag2v_spade_pair_workspace_floats(int N, int C, int HW);
int ag2v_spade_pair_stats(const float* xs, int N, int C, int HWs, int Ng, int rep, float eps, float momentum, float* mean, float* invstd,
                          float* rm0, float* rv0, float* rm1, float* rv1, float* workspace, void* stream);
int ag2v_spade_pair_apply(const float* xs, const float* mean, const float* invstd, const float* gb0, const float* gb1, float* y0, float* y1,
                          int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, void* stream);
/* ... with y0 / y1 stored as split words when split0 / split1 != 0 (ag2v_conv_opts::in_split of the convolutions that read them). */
int ag2v_spade_pair_apply_s(const float* xs, const float* mean, const float* invstd, const float* gb0, const float* gb1, float* y0,
                            float* y1, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, int split0, int split1,
                            void* stream);
int ag2v_spade_pair_bwd(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                        const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1, float* sums,
                        int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace, void* stream);
/* ... dgb_split is a flag word: 1 = dgb0 / dgb1 stored as split words (see ag2v_norm_bwd_gs); 2 = gb0 / gb1 point at compact
 * copies [N, C, H, W] of the gamma halves (the backward reads no beta; see ag2v_norm_bwd_gs mode 3). */
int ag2v_spade_pair_bwd_s(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                          const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1,
                          float* sums, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace,
                          int dgb_split, void* stream);
/* phases as ag2v_norm_bwd_gs_p; synchronised batch norm all-reduces columns 0, 1, 4, 5 of sums[N / Ng, C, 8] between them */
int ag2v_spade_pair_bwd_sp(const float* dy0, const float* y0, const float* dy1, const float* y1, const float* xs, const float* gb0,
                           const float* gb1, const float* mean, const float* invstd, float* dxs, float* dgb0, float* dgb1,
                           float* sums, int N, int C, int H, int W, int up, int Ng, float slope0, float slope1, float* workspace,
                           int dgb_split, int phase, float inv_count, void* stream);
int ag2v_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* invstd, int NC, int HW, float eps, float slope,
                            void* stream);
int ag2v_instnorm_lrelu_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                            float* dx, int NC, int HW, float slope, void* stream);

/* ---- layout render: boxes_to_layout models/layout.py:28-63 (+_boxes_to_grid :98-130, _pool_samples :205-237) -----
 * vecs[BT,O,D], boxes[BT,O,4] xywh, keep[BT,O] int32 (remove_dummy_objects models/utils.py:95-102) -> out[BT,D,H,W] */
int ag2v_layout_render_fwd(const float* vecs, const float* boxes, const int* keep, float* out, int BT, int O, int D, int H,
                           int W, void* stream);
int ag2v_layout_render_bwd(const float* dout, const float* boxes, const int* keep, float* dvecs, int BT, int O, int D, int H,
                           int W, void* stream);

/* layout convolution in low-rank form: conv(layout, Wt) with layout = sum_j v_j (x) w_j evaluated as
 * out = residual + sum_j sum_rs U[n,j,k,rs] * w_j(oy*st+r-pad, ox*st+s-pad),  U = Wt . v  (caller's GEMM);  R = 3 or 4 (square).
 * Replaces the layout channels of generator.py:74,82-83 (conv_dim_in) and flows_generator.py:32 (down_flow[0]). */
/* pool = 1: the convolution sees avg_pool2d(3,2,1,count_include_pad=False)(layout) (discriminator.py:270-271,351) */
int ag2v_layout_conv_fwd(const float* U, const float* boxes, const int* keep, const float* residual /*nullable*/, float* out,
                         int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, void* stream);
/* ag2v_layout_conv_fwd plus the IMAGE channels of the same convolution (the reference convolves cat[layouts, frame(s)],
 * generator.py:66,74,82-83): out += iscale[n] * conv(img[n, :IC], wimg[K, IC, R, R]) (iscale nullable; stride 1, pool 0, same-size
 * convolution, OW % 4 == 0, IC <= 8).  Their gradients are those of a plain convolution (ag2v_conv2d_dgrad_ex / _wgrad_ex on img).
 * J == 0 (U, boxes, keep may be NULL): no layout term -- out = residual + iscale[n] * conv(img, wimg), a dense convolution of <= 8
 * channels to K outputs on this kernel's 4-pixel x 16-channel FMA path: the data gradients of conv_img / conv_flow
 * (spade_generator.py:28,82, flows_generator.py:56), whose gathered tensor dy has 3 / 2 channels. */
int ag2v_layout_conv_img_fwd(const float* U, const float* boxes, const int* keep, const float* residual /*nullable*/, float* out,
                             int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, const float* img,
                             const float* wimg, const float* iscale /*nullable*/, int IC, void* stream);
/* ag2v_layout_conv_img_fwd that also writes the batch-norm statistics of `out` -- generator.py:30,66,74: conv_dim_in is followed by a
 * SynchronizedBatchNorm2d whose statistics pass would otherwise re-read the [NB, K, H, W] result -- as Chan partials (count, mean, M2)
 * per (sample, channel, band of 4 rows): stat_partial[NB / Ng][K][Ng * ceil(H / 4)][3] (ag2v_layout_conv_stats_floats floats), Ng = samples
 * per statistics group.  ag2v_bn_finalize_partials turns them into what ag2v_bn_stats_g returns.  Only for calls
 * ag2v_layout_conv_stats_supported() accepts (3 x 3, pad 1, stride 1, no pooling, W <= 256 in whole quads, K % 16 == 0, IC <= 3) with U and
 * wimg 16-byte aligned: AG2V_EINVAL otherwise. */
int ag2v_layout_conv_stats_supported(int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, int IC, int Ng);
size_t ag2v_layout_conv_stats_floats(int NB, int K, int H);
int ag2v_layout_conv_img_fwd_stats(const float* U, const float* boxes, const int* keep, const float* residual /*nullable*/, float* out,
                                   int NB, int J, int K, int H, int W, int R, int stride, int pad, int pool, const float* img,
                                   const float* wimg, const float* iscale /*nullable*/, int IC, float* stat_partial, int Ng, void* stream);
int ag2v_layout_conv_bwd(const float* dout, const float* boxes, const int* keep, float* dU, int NB, int J, int K, int H, int W,
                         int R, int stride, int pad, int pool, void* stream);

/* ---- flow warp: resample/get_grid models/utils.py:113-140 (grid_sample bilinear, border, align_corners=False) --- */
int ag2v_flow_warp_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream);
int ag2v_flow_warp_bwd(const float* dout, const float* img, const float* flow, float* dflow, float* dimg /*nullable*/,
                       int N, int C, int H, int W, void* stream);

/* ---- resampling: avg_pool2d(3,2,1,count_include_pad=False) discriminator.py:270-271; nearest F.interpolate /
 * nn.Upsample(scale_factor=2) normalization.py:102, spade_generator.py:36,55-70, flows_generator.py:47 ------------- */
int ag2v_avgpool3s2_fwd(const float* x, float* y, int NC, int H, int W, void* stream);
int ag2v_avgpool3s2_bwd(const float* dy, float* dx, int NC, int H, int W, void* stream);
/* max_pool2d(2, stride 2), floor mode: the pooling layers of VGG19.features behind the perceptual loss (architecture.py:93-123,
 * loss.py:102-118); backward routes each window's gradient to its first maximum (row-major), as ATen does. */
int ag2v_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, void* stream);
int ag2v_maxpool2_bwd(const float* dy, const float* x, float* dx, int NC, int H, int W, void* stream);
int ag2v_nearest_resize_fwd(const float* x, float* y, int NC, int H, int W, int OH, int OW, void* stream);
int ag2v_nearest_resize_bwd(const float* dy, float* dx, int NC, int H, int W, int OH, int OW, void* stream);
/* Adjoint of the whole nearest-neighbour pyramid of one segmentation map (every SPADE of a decoder pass interpolates the same map,
 * normalization.py:102): dx = g_full (nullable) + sum_l scatter(g_levels[l]); g_levels / hs / ws are host arrays of n_levels <= 8
 * entries (null level pointers are skipped), every level an integer down-sampling of H x W; W % 4 == 0. */
int ag2v_nearest_pyramid_bwd(const float* g_full, const float* const* g_levels, const int* hs, const int* ws, int n_levels,
                             float* dx, int NC, int H, int W, void* stream);

/* ---- action-graph convolution: GraphTripleConv.forward models/graph_models/graph.py:41-107 ---------------------
 * edges int64 [B,T,2]; valid uint8 [B,T] (pred_indicators).  gather: graph.py:64-68; pool: graph.py:79-100 */
int ag2v_triple_gather_fwd(const float* obj, const float* pred, const long long* edges, float* out, int B, int O, int T,
                           int Din, int Dp, void* stream);
int ag2v_triple_gather_bwd(const float* dtri, const long long* edges, float* dobj, float* dpred /*nullable*/, int B, int O,
                           int T, int Din, int Dp, void* stream);
int ag2v_triple_pool_fwd(const float* newt, const long long* edges, const unsigned char* valid, float* pooled, float* counts,
                         int B, int O, int T, int Hd, int Dp, void* stream);
int ag2v_triple_pool_bwd(const float* dpooled, const float* dp_in /*nullable*/, const long long* edges,
                         const unsigned char* valid, const float* counts, float* dnewt, int B, int O, int T, int Hd, int Dp,
                         void* stream);
/* nn.Linear / build_mlp (models/layers.py:6-25): strided fp32 GEMM  C = act(A*B + bias) */
int ag2v_gemm_f32(const float* a, const float* b, const float* bias, float* c, int M, int N, int K, long long sam,
                  long long sak, long long sbk, long long sbn, int ldc, int relu, int accumulate, void* stream);
int ag2v_relu_bwd(const float* dy, const float* y, float* g, long long n, void* stream);
int ag2v_col_sum(const float* x, float* out, int M, int N, void* stream);
/* nn.Embedding (models/attribute_embed.py:38-45, graph_models/model.py:143,154) */
int ag2v_embedding_fwd(const long long* idx, const float* table, float* out, int n, int D, int out_stride, int out_off,
                       void* stream);
int ag2v_embedding_bwd(const long long* idx, const float* dout, float* dtable, int n, int V, int D, int in_stride, int in_off,
                       void* stream);

/* ---- pointwise + losses: F.leaky_relu / tanh (spade_generator.py:78-80); GANLoss hinge networks/loss.py:57-76;
 * L1 feature matching / flow-warp loss_model.py:84,97 ------------------------------------------------------------- */
int ag2v_act_fwd(int op /*0 lrelu, 1 tanh*/, const float* x, float* y, long long n, float slope, void* stream);
int ag2v_act_bwd(int op, const float* dy, const float* y, float* dx, long long n, float slope, void* stream);
size_t ag2v_reduce_workspace_floats(void);
/* op 0: sum|a-b|  1: sum a  2: sum min(a-1,0)  3: sum min(-a-1,0);  out[0] = scale * sum */
int ag2v_reduce_fwd(int op, const float* a, const float* b, long long n, float scale, float* out, float* workspace,
                    void* stream);
int ag2v_reduce_bwd(int op, const float* a, const float* b, long long n, const float* gout, float scale, float* da,
                    void* stream);

/* ---- spectral norm: torch.nn.utils.spectral_norm forward pre-hook (call sites architecture.py:35-39,
 * flows_generator.py:79-85, normalization.py:26-27) ---------------------------------------------------------------- */
/* power_iter & 1: one in-place power iteration on (u, v) first (training mode); & 2 (AG2V_DETERMINISTIC): W^T u without row splits */
size_t ag2v_spectral_workspace_floats(int M);
int ag2v_spectral_norm_fwd(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, float* wbar,
                           float* sigma, float* workspace, void* stream);
/* as above, also writing copies of the u[M] / v[R] this call used (nullable) for the backward tape */
int ag2v_spectral_norm_fwd_save(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, float* wbar,
                                float* sigma, float* u_used, float* v_used, float* workspace, void* stream);
int ag2v_spectral_norm_bwd(const float* dwbar, const float* wbar, const float* u, const float* v, const float* sigma, int M,
                           int R, float* dw, float* workspace, void* stream);
/* accumulate != 0: dw += ... (weight_orig collects the gradients of all forward calls of a step in one buffer) */
int ag2v_spectral_norm_bwd_acc(const float* dwbar, const float* wbar, const float* u, const float* v, const float* sigma, int M,
                               int R, float* dw, int accumulate, float* workspace, void* stream);

/* Frame-batched form: the T train-mode forward calls of one spectral-norm convolution per step (one per generated frame) are T
 * consecutive power iterations on (u, v); sigma_t depends on W, u, v only, so all T are produced up front:
 *   sigma[T], inv_sigma[T] (= ag2v_conv_opts::gscale), u_used[T*M] / v_used[T*R] = the vectors behind sigma_t (nullable).
 *   power_iter == 0 (eval mode): T must be 1.  workspace: ag2v_spectral_workspace_floats(M). */
int ag2v_spectral_sigmas(const float* w, float* u, float* v, int M, int R, float eps, int power_iter, int T, float* sigma,
                         float* inv_sigma, float* u_used, float* v_used, float* workspace, void* stream);
/* Backward of y_n = scale[g(n)] * conv(x_n, W) from the per-group weight gradients G[T][n] (ag2v_conv2d_wgrad_ex, groups = T):
 * dw[i] (+)= sum_t scale[t] G[t][i];  dscale[t] = <G[t], W>.  accumulate: AG2V_ACCUMULATE | AG2V_DETERMINISTIC (one block, ordered). */
int ag2v_sn_group_combine(const float* G, const float* w, const float* scale, int T, long long n, float* dw, int accumulate,
                          float* dscale, void* stream);
/* Backward of scale[t] = 1 / (u_t^T W v_t):  dw[m, r] (+)= sum_t (-dscale[t] scale[t]^2) u_t[m] v_t[r]. */
int ag2v_spectral_sigmas_bwd(const float* dscale, const float* scale, const float* u_used, const float* v_used, int T, int M, int R,
                             float* dw, int accumulate, void* stream);

/* Multi-tensor form of ag2v_spectral_sigmas: every spectral-norm layer of a network per launch (3 launches per power iteration
 * for ALL layers).  descs: device array of n_layers records
 *   { const float* w; float* u; float* v; float* vtmp; float* wv; float* sigma; float* inv_sigma; float* u_used; float* v_used;
 *     int M; int R; int blk0; int row0; }          (ag2v_sn_desc_bytes() bytes each)
 * vtmp[R] / wv[M] are per-layer scratch (vtmp zero on entry, left zero on exit); blk0 / row0 = running sums over the layers of
 * ceil(R/256) * ceil(M/64) and of M; total_blocks / total_rows their totals.  Training mode only (T power iterations).
 * deterministic != 0: W^T u of a column block by ONE workgroup over all rows (stores, no atomics across row chunks). */
size_t ag2v_sn_desc_bytes(void);
int ag2v_spectral_sigmas_multi(const void* descs, int n_layers, int total_blocks, int total_rows, int T, float eps, int deterministic,
                               void* stream);

/* ---- graph-net programs: one persistent launch per recurrent pass of an action-graph net -------------------------------------
 * Replaces the per-operator launches of Acts2LayoutModel.forward's loop over time steps (models/graph_models/model.py:126-169),
 * GraphTripleConv (graph.py:41-107) and MultiscaleActionDiscriminator.get_obj_vecs (discriminator.py:273-313): the host writes the
 * pass as an array of operator records over caller-owned tensors and `grid` co-resident workgroups interpret it, meeting at a grid
 * barrier (agent-scope release / acquire) after every operator.  Record layout (ag2v_graph_op_bytes() = 192 bytes, little endian):
 *   int type, M, N, K;  const float *a, *b, *d, *bias, *res, *mask;  float *c;  const int *i0, *i1, *i2;  const unsigned char *valid;
 *   long long lda, ldb, ldd, ldr, ldm, ldc;  int w0, w1, w2, relu, acc, O, T, pad[3];
 * type 1 GEMM  c[m,n] = act(sum_k A'[m,k] b[n,k] + bias[n]) + res[m,n], A' = mask ? (mask[m,k] > 0 ? a : 0) : a  (exact fp32 MFMA)
 *      2 CAT   c[r,:] = [a[i0(r), :w0] | b[i1(r), :w1] | d[i2(r), :w2]]   (null index = r; null b / d = segment absent)
 *      3 POOL  masked average pooling onto objects (graph.py:79-100): a = new triples [B*T, 2 w0 + w1], i0 / i1 = subject / object
 *              ROW (b * O + object) per edge, valid per edge; c [B*O, w0]; counts written to `mask` (as floats)
 *      4 POOL_BWD  adjoint: a = dpooled, b = gradient of the predicate part (nullable), mask = counts; c [B*T, 2 w0 + w1]
 *      5 GATHER_BWD  c[bo, :w0] = res[bo] + sum over the sample's edges of the subject / object parts of a [B*T, 2 w0 + w1]
 *      6 TRANSPOSE  c[k, n] = a[n, k]        7 ADD3  c = a + b + d (b, d nullable), M rows of N columns
 *      POOL / GATHER_BWD take at most ag2v_graph_program_max_edges() edges per sample (T); every width, K and leading dimension is
 *      a multiple of 4 floats (16-byte accesses).  The record builder (ag2video_amd/graph_program.py) refuses anything else.
 * sync: two device words {arrival counter, timeout flag}, BOTH zeroed by the call; the flag is set when a grid barrier gave up (the
 * workgroups were not co-resident): the results of THAT run are invalid and the caller must read the flag before it trusts them. */
size_t ag2v_graph_op_bytes(void);
int ag2v_graph_program_grid(void);
int ag2v_graph_program_max_edges(void);
int ag2v_graph_program_run(const void* ops, int n_ops, unsigned* sync, int grid, void* stream);

/* ---- fused Adam over a flat fp32 buffer: torch.optim.Adam as configured at scripts/train.py:365-368 and
 * meta_models.py:70-72 (weight_decay 0, amsgrad off); step_index >= 1 --------------------------------------------- */
int ag2v_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   int step_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG2V_H */
