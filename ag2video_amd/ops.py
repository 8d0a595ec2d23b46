"""Host-side operators: torch.autograd.Function wrappers over the libag2v.so C ABI (include/ag2v.h).

PyTorch supplies device memory, the current HIP stream and the autograd tape; all arithmetic of these operators runs
in the hand-written gfx950 kernels of ag2video_amd/csrc.  Every operator requires CUDA(HIP) fp32 tensors and raises
otherwise -- there is no CPU path in the product (tests/ check these operators against oracle/ on the GPU).

Side channels between operators travel as attributes on the tensor objects autograd hands from one Function to the next (the
same Python object arrives; a tensor that lost its attribute simply takes the plain path):
  _ag2v_split          the tensor holds split words (bf16 hi | bf16 lo per value) instead of fp32: is_split / mark_split;
  _ag2v_dy_split_ok    on an operator's OUTPUT: its backward accepts the gradient of this tensor as split words (a _DySplitOffer);
  _ag2v_channel_sums   on a GRADIENT tensor: its per-channel sums, already computed by the kernel that produced it (the bias
                       gradient of the convolution it goes to).
These only make sense for a tensor with ONE consumer (autograd would add a second gradient contribution to split words as if they
were floats).  That is enforced, not assumed: the attributes carry the tensor's version counter (an in-place accumulation into the
tensor makes is_split raise and voids the channel sums), an offer can be taken once (a second taker raises in the forward pass),
and the producer's backward raises when a taken offer's gradient arrives as anything but the taker's split-word tensor.
"""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _lib

_P = _lib.c_void_p


def _ptr(t):
    return None if t is None else _P(t.data_ptr())


def _stream():
    return _P(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("ag2video_amd.ops: expected a HIP device tensor, got %s (no CPU fallback)" % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError("ag2video_amd.ops: expected float32, got %s" % t.dtype)


def _c(t):
    return None if t is None else t.contiguous()


def _ptr_off(t, elems):
    return _P(t.data_ptr() + 4 * elems)


# ---- synchronised batch norm on every norm path (reference: sync_batchnorm/batchnorm.py:80-93 forward, :105-126 reduce / broadcast,
# :128-145 _compute_mean_std).  Forward: the local (mean, inv_std) of a statistics kernel -> ONE all-reduce of [sum, sum of squares]
# -> the reference's multi-device formula (inv_std = clamp(biased var, eps)^-1/2, running_var fed with the unbiased variance).
# Backward: every fused backward is a reduce pass and an apply pass (the *_p entry points); between them (sum dxhat, sum dxhat xhat)
# are summed over the ranks, everything else in `sums` (affine / bias gradients) stays per rank.
def _sync_world():
    from . import distributed as D
    return D.sync_bn()


@torch.no_grad()
def _sync_combine(mean_l, invstd_l, groups, C, count, eps, momentum, running_pairs):
    """Local statistics [groups * C] over `count` elements per (group, channel) -> (global mean, inv_std, world); the running buffers
    of `running_pairs` get the per-group updates, in order, in closed form."""
    from . import distributed as D
    var_l = invstd_l.pow(-2) - eps
    mean, invstd, unb, cnt = D.combine_bn_stats(mean_l.view(groups, C), var_l.view(groups, C), count, eps)
    world = int(round(cnt / count))
    k = torch.arange(groups - 1, -1, -1, device=mean_l.device, dtype=torch.float32)
    coef = momentum * (1.0 - momentum) ** k                                   # weight of group g in the final value
    keep = (1.0 - momentum) ** groups
    for rm, rv in running_pairs:
        if rm is not None:
            rm.mul_(keep).add_((coef.view(-1, 1) * mean).sum(0))
            rv.mul_(keep).add_((coef.view(-1, 1) * unb).sum(0))
    return mean.reshape(-1).contiguous(), invstd.reshape(-1).contiguous(), world


@torch.no_grad()
def _sync_bwd_sums(sums, n_norms):
    """sums [G, C, 4 * n_norms] of a reduce pass: columns 4 k + (0, 1) summed over the ranks in place; returns this rank's own copy."""
    from . import distributed as D
    local = sums.clone()
    idx = [4 * k + i for k in range(n_norms) for i in (0, 1)]
    part = sums[..., idx].contiguous()
    D.sum_over_ranks_(part)
    sums[..., idx] = part
    return local


_PRECISIONS = {'fp32': 0, 'bf16x3': 1}
_PRECISION = [_PRECISIONS.get(os.environ.get('AG2V_CONV_PRECISION', 'bf16x3'), 1)]


def set_conv_precision(name):
    """'bf16x3' (default): conv GEMMs with enough rows multiply 3-term bf16 splits on the bf16 MFMA pipe with fp32 accumulation
    (fp32-level accuracy, ~1.6-2x the exact path); 'fp32': exact fp32 MFMA everywhere.  The value is handed to the library with
    every call (ag2v_conv_opts.precision); nothing process-wide changes in the library."""
    _PRECISION[0] = _PRECISIONS[name]


def get_conv_precision():
    return {v: k for k, v in _PRECISIONS.items()}[_PRECISION[0]]


@contextlib.contextmanager
def conv_precision(name):
    """set_conv_precision for the duration of a with block."""
    old = _PRECISION[0]
    _PRECISION[0] = _PRECISIONS[name]
    try:
        yield
    finally:
        _PRECISION[0] = old


# ---- deterministic reductions (ag2v_conv_opts::deterministic, AG2V_DETERMINISTIC).  Default off: split-K / split-pixel partial sums
# meet in fp32 atomics, whose order -- and therefore the last bits of every sum -- changes from run to run.  On: forward / data
# gradient never split a tile's reduction, the weight gradient writes its partials to a workspace and adds them in order, the
# bias-gradient sums, the spectral-norm power iteration and the per-group scale gradients run without atomics.  Two runs of the same
# step are then bit-identical (tests/test_fullsize_gpu.py).  One exception: the image gradient of flow_warp (scatter-add; only needed
# when the previous frame is a generated one: test_mode / bp_prev, not the teacher-forced training step).
_DETERMINISTIC = [os.environ.get('AG2V_DETERMINISTIC', '0') not in ('', '0')]


def set_deterministic(flag):
    _DETERMINISTIC[0] = bool(flag)


def is_deterministic():
    return _DETERMINISTIC[0]


# Lean tape (AG2V_LEAN_TAPE=1 / set_lean_tape / --lean_tape 1): trade a few passes over memory for tape size.
#  * SPADE modulations keep a compact copy of gamma for their backward instead of the whole gamma | beta tensor (the backward never
#    reads beta: half of every [N, 2C, H, W] tensor of the tape is dead weight);
#  * the batch norm behind the low-rank layout convolution does not keep its input: the backward runs that convolution's last kernel
#    again (deterministic, so the recomputed tensor is bit-identical) when everything else of the tape has been freed.
# c3: 138.6 -> ~105 GB peak for ~ +1.5 % step time.  Results are bit-identical to the default mode's in deterministic mode.
_LEAN_TAPE = [os.environ.get('AG2V_LEAN_TAPE', '0') not in ('', '0')]


def set_lean_tape(flag):
    _LEAN_TAPE[0] = bool(flag)


def lean_tape():
    return _LEAN_TAPE[0]


class deterministic(object):
    """Context manager: operators called inside use the deterministic (atomic-free, ordered) reductions; the flag travels with
    every call (per-call options of the C ABI), nothing process-wide changes in the library."""

    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        self.old = _DETERMINISTIC[0]
        _DETERMINISTIC[0] = self.flag
        return self

    def __exit__(self, *exc):
        _DETERMINISTIC[0] = self.old
        return False


def _det_bit():
    return 2 if _DETERMINISTIC[0] else 0


# ---- optional per-launch timing of the convolution kernels (bench.py's roofline leg): HIP events on the launch stream
_TIMING = None


def enable_kernel_timing(flag=True):
    """Start (True) / stop (False) recording a HIP event pair around every convolution kernel launch.  Starting clears the
    accumulated records."""
    global _TIMING
    if flag:
        _FOLDED.clear()
    _TIMING = [] if flag else None


# ---- parameter epoch: bumped by the fused optimizer (its kernels update parameters through raw pointers, which does not
# advance torch's per-tensor version counters); caches of tensors derived from parameters key on it.
_PARAM_EPOCH = [0]


def bump_param_epoch():
    _PARAM_EPOCH[0] += 1


def param_epoch():
    return _PARAM_EPOCH[0]


# ---- which GEMMs take the bf16x3 tiles.  The library default (128 rows, AG2V_BF16X3_MIN_M) is the conservative policy; the
# model code lowers it for the image decoder and the discriminator, and keeps 128 inside the flow network: there the
# 64-channel full-resolution tail feeds the warp, which amplifies its 1e-5 split error ~100x on the reference golden toy
# (imgs_pred 1.3e-3 against a 1e-3 tolerance), everywhere else the error stays at 1e-5.
_MIN_ROWS_STACK = []
# the row threshold of a convolution created outside any bf16x3_min_rows block.  Host-side: it travels with every call
# (ag2v_conv_opts::min_rows), the library's per-device default is never consulted
DEFAULT_MIN_ROWS = max(int(os.environ.get('AG2V_BF16X3_MIN_M', '0') or 0), 0) or 128


class bf16x3_min_rows(object):
    """Context manager: convolutions *created* inside send GEMMs with at least ``rows`` rows to the bf16x3 tiles in their forward
    pass and with at least ``bwd_rows`` (default: the same) in their backward pass (data / weight gradient)."""

    def __init__(self, rows, bwd_rows=None, fwd_h16=False):
        # fwd_h16: the FORWARD launches that take the split tiles use the fp16 3-term split (ag2v_conv_opts::precision = 2: ~5e-6,
        # fp32-class, for normalised operands -- the flow network, whose output the warp amplifies) instead of the bf16 one
        self.rows = (int(rows), int(rows if bwd_rows is None else bwd_rows), bool(fwd_h16))

    def __enter__(self):
        _MIN_ROWS_STACK.append(self.rows)
        return self

    def __exit__(self, *exc):
        _MIN_ROWS_STACK.pop()
        return False


EXACT = 1 << 30      # a row threshold no GEMM reaches: exact fp32 MFMA

# ---- frame groups.  In training the generator's frame loop is teacher-forced (generator.py:71): its T = F - n_frames_G + 1 passes
# are data-independent and run here as ONE pass over a frame-major [T*B, ...] batch.  What the reference computes per PASS stays
# per frame group: batch-norm statistics (+ one running-stat update per pass, in order) and spectral norm's sigma_t (+ one power
# iteration per pass).  Modules read the group count from this context (SynchronizedBatchNorm2d, SpectralConv2d).
_FRAME_GROUPS = [1]


class frame_groups(object):
    def __init__(self, T):
        self.T = int(T)

    def __enter__(self):
        _FRAME_GROUPS.append(self.T)
        return self

    def __exit__(self, *exc):
        _FRAME_GROUPS.pop()
        return False


def current_frame_groups():
    return _FRAME_GROUPS[-1]



def _current_min_rows():
    """(forward threshold, backward threshold), or (None, None) = library default."""
    return _MIN_ROWS_STACK[-1][:2] if _MIN_ROWS_STACK else (None, None)


def _current_fwd_h16():
    return bool(_MIN_ROWS_STACK and _MIN_ROWS_STACK[-1][2] and _PRECISION[0] == 1)


def _conv_label(mode, M, min_rows=None):
    """Placeholder label; _launch replaces conv labels by the name of the kernel the library actually launched."""
    return "conv_%s" % mode


_DET_WS_FLOATS = {}


def reload_kernel_switches():
    """Tests / tools: after changing one of the library's diagnostic environment switches (AG2V_NO_HALO, AG2V_NO_WGRAD_HALO,
    AG2V_HALO_MIN_TILES, AG2V_HALO_MFMA32), with no launch in flight -- the library re-reads them (it caches them: getenv beside a
    concurrent putenv is not thread-safe) and the host forgets the workspace sizes it has cached per launch shape."""
    _lib.call('ag2v_halo_reload_env')
    _DET_WS_FLOATS.clear()


def _conv_launch(min_rows, name, label, flops, *args, shape=None, gscale=None, group_n=1, in_split=False, dy_split=False,
                 out_split=False, det_ws_query=None, precision=None, out_nstride=0, out_mstride=0, chan_sums=None):
    """One ag2v_conv2d_*_ex call.  Arithmetic (precision, bf16x3 row threshold: ``min_rows``, None = DEFAULT_MIN_ROWS) and the
    per-sample-group output scale travel as per-call options (ag2v_conv_opts): no process-wide switch is touched.
    Returns ag2v_conv_opts::chan_sums_done (data gradient with ``chan_sums``: the launch added the channel sums of dx to it)."""
    opts = _lib.ConvOpts(_PRECISION[0] if precision is None else int(precision), int(min_rows) if min_rows else DEFAULT_MIN_ROWS, None if gscale is None else gscale.data_ptr(),
                         int(group_n), None, 1 if in_split else 0, 1 if dy_split else 0, 1 if out_split else 0,
                         1 if _DETERMINISTIC[0] else 0, None, 0, int(out_nstride), int(out_mstride),
                         None if chan_sums is None else chan_sums.data_ptr(), 0)
    det_ws = None
    if det_ws_query is not None:      # weight gradient: slabs for the ordered reduction of its pixel-range splits (deterministic mode;
        # in either mode for the LDS-DMA kernel, whose atomic epilogue costs 6-16 % of a launch)
        dev, qargs = det_ws_query
        # (the size depends on the shape and on the per-call options only: one library query per distinct launch)
        qkey = qargs + (opts.precision, opts.min_rows, opts.in_split, opts.dy_split, opts.deterministic)
        n = _DET_WS_FLOATS.get(qkey)
        if n is None:
            n = _DET_WS_FLOATS[qkey] = int(_lib.query('ag2v_conv2d_wgrad_det_workspace_floats', *qargs, ctypes.byref(opts)))
        if n:
            det_ws = torch.empty(n, device=dev, dtype=torch.float32)
            opts.det_workspace, opts.det_workspace_floats = det_ws.data_ptr(), n
    args = args + (ctypes.byref(opts),)
    if _TIMING is None or _PAUSED:
        _lib.call(name, *args)
        return bool(opts.chan_sums_done)
    e0 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
    e1 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    # key the record on the HIP kernel the library chose (tile shape / arithmetic / load path), as rocprofv3 names it; the
    # pass (fwd / dgrad / wgrad) stays in front because forward and data gradient share the implicit-GEMM kernels
    _TIMING.append(("%s:%s" % (label, (opts.kernel or b'?').decode()), flops, e0, e1, shape, _SCOPE[-1]))
    return bool(opts.chan_sums_done)


_PAUSED = False

# ---- timing scopes: which part of the model a convolution call belongs to ('netG' = the SPADE decoder stack north_star's MFMA target
# is stated on, 'flow', 'conv_dim_in', 'D', ...).  Modules open a scope around their forward; the convolution nodes remember it for their
# backward launches.  Only the timing records read it.
_SCOPE = ['other']


class timing_scope(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _SCOPE.append(self.name if self.name is not None else _SCOPE[-1])
        return self

    def __exit__(self, *exc):
        _SCOPE.pop()
        return False


def _scoped_backward(fn):
    def backward(ctx, *grads):
        with timing_scope(getattr(ctx, 'scope', None)):
            return fn(ctx, *grads)
    return backward


def pause_kernel_timing(flag):
    """Temporarily stop (True) / resume (False) recording without dropping what has been recorded."""
    global _PAUSED
    _PAUSED = bool(flag)


def _launch(name, label, flops, *args, shape=None):
    if _TIMING is None or _PAUSED:
        _lib.call(name, *args)
        return
    e0 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
    e1 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    _TIMING.append((label, flops, e0, e1, shape, _SCOPE[-1]))


_EVENT_POOL = []      # HIP events are recycled across steps: creating ~20k of them inside a timed step costs ~2 %
_FOLDED = {}          # records already converted to milliseconds: (label, shape, scope) -> [launches, flops, ms]


def fold_kernel_timing(keep=True):
    """Synchronise, convert the event pairs recorded so far into accumulated milliseconds (keep=False: discard them, e.g. after a
    warm-up step) and return the events to the pool.  bench.py calls it between timed steps so that a step's events are reused by
    the next one."""
    torch.cuda.synchronize()
    if not _TIMING:
        return
    for label, flops, e0, e1, shape, scope in _TIMING:
        if keep:
            d = _FOLDED.setdefault((label, shape, scope), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += e0.elapsed_time(e1)
        _EVENT_POOL.append(e0)
        _EVENT_POOL.append(e1)
    del _TIMING[:]


def kernel_timing_summary(by_shape=False, by_scope=False):
    """{label: {launches, flops, ms}} for everything recorded since enable_kernel_timing(True); synchronises.  by_scope: keys are
    "<scope>|<label>" (timing_scope: 'netG', 'flow', 'conv_dim_in', 'D', 'other')."""
    fold_kernel_timing(True)
    out = {}
    for (label, shape, scope), (n, flops, ms) in _FOLDED.items():
        if by_scope:
            label = "%s|%s" % (scope, label)
        if by_shape:
            label = "%s %s" % (label, shape)
        d = out.setdefault(label, {'launches': 0, 'flops': 0.0, 'ms': 0.0})
        d['launches'] += n
        d['flops'] += flops
        d['ms'] += ms
    return out


# ---- weight-gradient sinks.  A weight tensor used by several convolution calls of one graph (the generator runs once per frame:
# 15 uses of every weight per step) gets ONE gradient buffer per backward pass: the first weight-gradient launch of the pass
# writes it and hands it to autograd, the later ones add into it in place (ag2v_conv2d_wgrad_acc) and hand autograd nothing.
# Autograd delivers the buffer to the weight's producer (cat / spectral norm / AccumulateGrad) only after all of the weight's
# consumers have run, so it arrives complete; without this it would receive 15 tensors and sum them pairwise (4 700 small add
# launches and as many zero-fills per step).  The registry is cleared by an engine callback at the end of every backward pass.

# The same holds for spectral norm's weight_orig (one _SpectralWeight node per frame pass: 560 full-size adds per step without it)
# and for convolution biases.
_WSINK = {}          # id(weight tensor) -> [weakref to it, sink tensor or None]
_EXTRA_SINKS = True
_MERGE_SHARED_DGRAD = True      # conv2d_shared_input: one data-gradient launch
_MERGE_SHARED_WGRAD = True      # ... and one weight-gradient launch
_GRAD_SLOTS = os.environ.get('AG2V_NO_GRAD_SLOTS') is None                # ... whose concatenated dy the consumers write in place (_GradSlots; A/B)
_SLOTS_MIN_PIXELS = 65536  # ... on maps of at least this many pixels per batch (256 pixel tiles)
_SWAP_FEW_WGRAD = True         # <= 4-row weight gradients with swapped operands
_SWAP_WIDE_WGRAD = True   # ... and layers with fewer output than input channels
# <= 4-row same-size weight gradients on the library's one-workgroup-per-input-channel kernel (conv_wgrad_fewrow_kernel) instead of the
# swapped GEMM: OFF -- on 256 x 256 maps its dword gathers lose to the GEMM (tools/fewrow_wgrad_bench.py: conv_img 3.17 vs 1.50 ms,
# conv_flow 1.40 vs 1.26); the kernel serves the layers the swap does not cover (the discriminators' 4 x 4 last layer: 1.15 -> 0.35 ms)
_FEWROW_WGRAD_DIRECT = False
_PAD_WGRAD = True                # odd-width stride-1 weight gradients on the 16-byte-load kernel via zero columns
_FEWIN_DGRAD = True              # data gradients of <= 4-row layers on the layout kernel's image path
_WSINK_CB = [None]   # graph-task id of the backward pass whose end-of-pass callback is queued (None: none; ids start at 0)


def _wsink_register(w):
    e = _WSINK.get(id(w))
    if e is None or e[0]() is not w:
        _WSINK[id(w)] = [weakref.ref(w), None, -1]
    return id(w)


_DIRECT_STREAMS = []  # HIP streams on which launches of the current backward pass wrote straight into a leaf's .grad


def _wsink_clear():
    """End-of-pass callback (runs on the thread that called backward, after the last node).  Direct .grad writes are invisible to
    autograd's own leaf-stream synchronisation, so the caller's stream is made to wait here for every other stream that performed
    one (nodes run on the stream of their forward: the side-stream graph nets) -- an optimizer step or all-reduce issued after
    backward() then sees complete gradients without the caller having to know which streams were involved."""
    _WSINK.clear()
    _WSINK_CB[0] = None
    if _DIRECT_STREAMS and _JOIN_AT_END[0]:
        join_direct_grad_streams()


def join_direct_grad_streams():
    """Make the current stream wait for every other stream on which a backward pass wrote parameter gradients directly (see
    _wsink_clear).  Called from the end-of-pass callback, or -- accumulate_into_param_grads(join_streams=False) -- by the caller
    itself right before it reads the gradients (engine.gd_step: before Adam(G), after the discriminator phase has been queued)."""
    if _DIRECT_STREAMS:
        cur = torch.cuda.current_stream()
        for s_ in _DIRECT_STREAMS:
            if s_ != cur:
                cur.wait_stream(s_)
        del _DIRECT_STREAMS[:]


def _wsink_begin(task):
    if _WSINK_CB[0] != task:            # first sink use of this backward pass
        if _WSINK_CB[0] is not None:    # an earlier pass never reached its callback: release its buffers
            for ent in _WSINK.values():
                if ent[2] != task:
                    ent[1] = None
                    del ent[3:]
        torch.autograd.Variable._execution_engine.queue_callback(_wsink_clear)
        _WSINK_CB[0] = task


_DIRECT_GRAD = True
_USE_CHANNEL_SUMS = True
_SPLIT_NORM_DX = os.environ.get('AG2V_NO_SPLIT_NORM_DX') is None           # SPADE backwards write d(x) as split words for x's producer (A/B)
_DIRECT_DEPTH = [0]
_JOIN_AT_END = [True]


class accumulate_into_param_grads(object):
    """Context manager around a ``loss.backward()`` that ACCUMULATES into the parameters' .grad (the training step): inside it the
    weight / bias gradient launches of leaf parameters whose .grad is preallocated (the flat-buffer optimizers) add straight into
    .grad and hand autograd nothing.  Outside it -- torch.autograd.grad(...), backward(inputs=...), any caller that has not said so --
    every gradient is returned to autograd, which delivers it the usual way (the direct path returned None to
    autograd.grad and still wrote .grad)."""

    def __init__(self, join_streams=True):
        # join_streams=False: the caller joins the writing streams itself (ops.join_direct_grad_streams()) before it reads .grad --
        # lets side-stream backward work (the graph nets) run under whatever the caller queues next on its own stream
        self.join = bool(join_streams)

    def __enter__(self):
        _DIRECT_DEPTH[0] += 1
        self.old_join = _JOIN_AT_END[0]
        _JOIN_AT_END[0] = self.join
        return self

    def __exit__(self, *exc):
        _DIRECT_DEPTH[0] -= 1
        _JOIN_AT_END[0] = self.old_join
        return False


def _wsink_take(key, like):
    """-> (buffer to pass to the wgrad launch, accumulate flag, tensor to return to autograd or None).
    A sink buffer belongs to ONE backward pass: it carries the autograd graph-task id it was created under, so a buffer left
    behind by a pass that ended without its end-of-pass callback (an exception inside backward) is never accumulated into.
    Leaf parameters whose .grad is already allocated (the flat-buffer optimizers keep every .grad as a zeroed view of one
    gradient buffer) take the short way: the launch accumulates straight into .grad and autograd is handed nothing -- what
    AccumulateGrad would have done with one more buffer and one more add per parameter and backward pass."""
    e = _WSINK.get(key)
    task = torch._C._current_graph_task_id()
    if e is None or task < 0:           # registered before an earlier backward pass cleared the registry: plain path
        buf = torch.empty_like(like)
        return buf, 0, buf
    if _DIRECT_GRAD and _DIRECT_DEPTH[0] > 0:
        p = e[0]()
        if (p is not None and p.is_leaf and p.grad is not None and p.grad.shape == like.shape and p.grad.is_contiguous()
                and p.grad.dtype == torch.float32 and p.grad.device == like.device and not p._backward_hooks
                and not getattr(p, '_post_accumulate_grad_hooks', None)):
            _wsink_begin(task)              # (queues the end-of-pass callback, which joins the streams recorded here)
            st = torch.cuda.current_stream()
            if st not in _DIRECT_STREAMS:
                _DIRECT_STREAMS.append(st)
            return p.grad, 1, None
    _wsink_begin(task)
    if e[1] is None or e[2] != task:
        e[1], e[2] = torch.empty_like(like), task
        return e[1], 0, e[1]
    return e[1], 1, None


def _wsink_take_group(keys, likes):
    """Sinks of several weights as consecutive slices of one buffer -> (buffer, accumulate flag, tensors for autograd).
    Falls back to a fresh buffer (plain path) when any of the weights has no live registry entry."""
    task = torch._C._current_graph_task_id()
    ents = [_WSINK.get(k) for k in keys]
    rows = [t.shape[0] for t in likes]
    shape = (sum(rows),) + tuple(likes[0].shape[1:])
    if task < 0 or any(e is None for e in ents):
        cat = torch.empty(shape, device=likes[0].device, dtype=torch.float32)
        return cat, 0, list(torch.split(cat, rows, dim=0))
    _wsink_begin(task)
    e0 = ents[0]
    cat = e0[3] if (len(e0) > 3 and e0[2] == task) else None
    if cat is not None and tuple(cat.shape) == shape and all(e[2] == task for e in ents):
        return cat, 1, [None] * len(keys)
    cat = torch.empty(shape, device=likes[0].device, dtype=torch.float32)
    views = list(torch.split(cat, rows, dim=0))
    for e, v in zip(ents, views):
        e[1], e[2] = v, task
    del e0[3:]
    e0.append(cat)
    return cat, 0, views


# ------------------------------------------------------------------------------------------------ convolution
def _conv_fwd_launch(mr, x, w, bias, residual, y, stride, pad, relu, gscale=None, group_n=1, in_split=False, out_split=False, out_mstride=0):
    """out_mstride > 0: y is [n, N, 128, OH, OW] -- every 128 of w's n * 128 output channels form their own dense tensor
    (ag2v_conv_opts::out_mstride; the caller has asked _fwd_mtile_ok)."""
    N, C, H, W = x.shape
    h16 = _current_fwd_h16() and not in_split and not out_split
    K, _, R, S = w.shape
    ws = torch.empty(_lib.query('ag2v_conv2d_workspace_floats', C, K, R, S), device=x.device, dtype=torch.float32)
    _conv_launch(mr, 'ag2v_conv2d_fwd_ex', 'conv_fwd', 2.0 * N * K * y.shape[-2] * y.shape[-1] * C * R * S,
                 _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(y), N, C, H, W, K, R, S, stride, pad, 1 if relu else 0,
                 _ptr(ws), _stream(), shape=(N, C, H, W, K, R, stride), gscale=gscale, group_n=group_n, in_split=in_split,
                 out_split=out_split, precision=(2 if h16 else None), out_mstride=out_mstride)


_MERGE_SHARED_FWD = os.environ.get('AG2V_NO_MERGED_FWD') is None      # conv2d_shared_input: one forward launch (tests compare both sides)
# conv2d(relu_input=True) backward: the data-gradient launch adds dx's channel sums -- the bias gradient of the convolution behind the
# ReLU -- in its epilogue (ag2v_conv_opts::out_chan_sums) instead of a pass over dx by the producer's backward (tests compare both sides)
_CHAN_SUMS_IN_DGRAD = os.environ.get('AG2V_NO_DGRAD_SUMS') is None


def _fwd_mtile_ok(mr, x, K, R, S, stride, pad, in_split, out_split):
    N, C, H, W = x.shape
    h16 = _current_fwd_h16() and not in_split and not out_split
    return bool(_lib.query('ag2v_conv2d_fwd_mtile_ok', N, C, H, W, int(K), int(R), int(S), int(stride), int(pad), int(mr) if mr else DEFAULT_MIN_ROWS,
                           2 if h16 else _PRECISION[0], 1 if in_split else 0, 1 if out_split else 0, 1 if _DETERMINISTIC[0] else 0))


def _conv_dgrad_launch(mr, dy, w, mask, dx, stride, pad, acc, gscale=None, group_n=1, dy_split=False, out_split=False, chan_sums=None):
    """dx (+)= data gradient.  The workspace holds the re-ordered filters and, for <= 4 input channels, the intermediate of the
    1x1-GEMM form (caller memory: the library allocates nothing).  dx may be a channel slice of a wider [N, Ctot, H, W] buffer
    (_GradSlots): the kernels then step Ctot * H * W floats between samples (ag2v_conv_opts::out_nstride).
    chan_sums ([C] floats, initialised by the caller): -> True when the launch ADDED dx's per-channel sums to it from its epilogue
    (ag2v_conv_opts::out_chan_sums: the LDS-DMA kernels outside deterministic mode); False: untouched, the caller sums dx."""
    N, C, H, W = dx.shape
    onst = 0
    if not dx.is_contiguous():
        if dx.stride()[1:] != (H * W, W, 1) or dx.stride(0) < C * H * W:
            raise RuntimeError("conv2d data gradient: dx must be dense or a channel slice of a dense buffer, got strides %s" % (dx.stride(),))
        onst = dx.stride(0)
    K, _, R, S = w.shape
    nws = _lib.query('ag2v_conv2d_dgrad_workspace_floats', N, C, H, W, K, R, S, stride, pad)
    ws = torch.empty(nws, device=dy.device, dtype=torch.float32)
    return _conv_launch(mr, 'ag2v_conv2d_dgrad_ex', 'conv_dgrad', 2.0 * N * K * dy.shape[2] * dy.shape[3] * C * R * S,
                        _ptr(dy), _ptr(w), _ptr(mask), _ptr(dx), N, C, H, W, K, R, S, stride, pad, 1 if acc else 0, _ptr(ws), nws, _stream(),
                        shape=(N, C, H, W, K, R, stride), gscale=gscale, group_n=group_n, dy_split=dy_split, out_split=out_split,
                        out_nstride=onst, chan_sums=chan_sums)


def _conv_wgrad_launch(mr, dy, x, dw, K, R, S, stride, pad, acc, groups=1, in_split=False, dy_split=False, workspace=True):
    """dw (+)= weight gradient; groups > 1: dw is [groups, K, C, R, S], one gradient per group of N / groups samples.
    workspace=False (tests): no slab workspace is handed over -- the pixel-range splits then meet in fp32 atomics (the C ABI's
    behaviour for a caller that passes none; refused in deterministic mode)."""
    N, C, H, W = x.shape
    _conv_launch(mr, 'ag2v_conv2d_wgrad_ex', 'conv_wgrad', 2.0 * N * K * dy.shape[2] * dy.shape[3] * C * R * S,
                 _ptr(dy), _ptr(x), _ptr(dw), N, C, H, W, K, R, S, stride, pad, 1 if acc else 0, int(groups), _stream(),
                 shape=(N, C, H, W, K, R, stride), in_split=in_split, dy_split=dy_split,
                 det_ws_query=(x.device, (N, C, H, W, K, R, S, stride, pad, int(groups))) if workspace else None)


def _wgrad_tile_area(rows, cols):
    """Padded GEMM area of a bf16x3 weight-gradient launch with `rows` dense rows and `cols` gathered columns (plan_wgrad_bf16x3:
    256 x 128 tiles from 192 rows on when they pad no more rows, else 128 x 256)."""
    tall = rows >= 192 and -(-rows // 256) * 256 <= -(-rows // 128) * 128
    bm, bn = (256, 128) if tall else (128, 256)
    return (-(-rows // bm) * bm) * (-(-cols // bn) * bn)


def _wgrad_swap_pays(mr, x_shape, dy_shape, K, R, S, stride, pad, x_split, dy_split):
    """A same-size stride-1 weight gradient may run with its operands swapped: x becomes the dense row operand, dy the gathered
    one, and the launch computes dW'[c][k][r'][s'] = dW[k][c][R-1-r'][S-1-s'].  The kernels are bound by their producer waves
    (gather + split of the column operand), whose work and the matrix work both follow the PADDED tile area: a layer with fewer output
    than input channels (up_3's 128 -> 64 and the flow network's 64 -> 32 layers: 64 / 32 rows on a 128-row tile) pads less the
    other way round -- 6.0 -> 3.6 ms for 120 x 128 x 256 x 256 -> 64."""
    N, C, H, W = x_shape
    if not (_SWAP_WIDE_WGRAD and K > 4 and stride == 1 and R == S and 2 * pad == R - 1 and tuple(dy_shape[2:]) == (H, W) and _PRECISION[0] == 1):
        return False
    if _wgrad_tile_area(C, K * R * S) > 0.8 * _wgrad_tile_area(K, C * R * S):
        return False
    # the swapped launch reads dy in x's place (and the other way round): each must be readable in that role
    # (mr: the node's backward row threshold, recorded in its forward)
    if (mr or DEFAULT_MIN_ROWS) > min(C, K):
        return False
    if dy_split and not _lib.query('ag2v_conv2d_split_input_ok', N, K, H, W, C, R, S, 1, pad, int(mr or 0), int(mr or 0), 1):
        return False
    if x_split and not _lib.query('ag2v_conv2d_split_dy_ok', N, K, H, W, C, R, S, 1, pad, int(mr or 0), 1):
        return False
    return True


def _wgrad_pad_pays(mr, x_shape, dy_shape, K, R, S, stride, pad, x_split, dy_split):
    """A stride-1 bf16x3 weight gradient on fp32 operands whose output rows are not multiples of 8 pixels, large enough for the two
    padded copies to pay (see the call site)."""
    N, C, H, W = x_shape
    OH, OW = int(dy_shape[2]), int(dy_shape[3])
    if not (_PAD_WGRAD and stride == 1 and OW % 8 != 0 and OW >= 16 and not x_split and not dy_split and _PRECISION[0] == 1 and pad <= 3):
        return False
    if (mr or DEFAULT_MIN_ROWS) > K or K < 128 or C < 64:
        return False
    return (-OW % 8) * 5 <= OW and N * OH * OW >= 16384       # at most 20 % more columns


def _conv_wgrad_swapped(mr, dy, x, out, K, R, S, pad, acc, groups, x_split, dy_split):
    """out (+)= the weight gradient [groups x] [K, C, R, S], computed as the swapped launch's [groups x] [C, K, R, S] (see above)."""
    C = x.shape[1]
    t = torch.empty(((groups,) if groups > 1 else ()) + (C, K, R, S), device=dy.device, dtype=torch.float32)
    _conv_wgrad_launch(mr, x, dy, t, C, R, S, 1, pad, 0, groups=groups, in_split=dy_split, dy_split=x_split)
    t = t.flip(-2, -1).transpose(-4, -3)
    if acc:
        out.add_(t)
    else:
        out.copy_(t)


class _Conv2d(torch.autograd.Function):
    """relu: y = max(conv, 0) in the forward epilogue; its backward masks dy with (y > 0) unless ``grad_premasked`` says the
    consumer of y already did (see relu_input).  relu_input: x is itself a ReLU output whose only consumer is this convolution
    and whose producer was called with grad_premasked=True: the data gradient is returned as (x > 0) ? dx : 0, i.e. the ReLU's
    backward rides in this convolution's dgrad epilogue instead of a separate pass over dx.
    gscale [G] (optional): y_n = gscale[n // (N / G)] * conv(x_n, w) + bias + residual -- a spectral-norm convolution of the
    frame-batched generator (w = weight_orig, gscale = 1 / sigma_t of frame group t: spectral_scales)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, gscale, stride, pad, relu, grad_premasked, relu_input, x_split=False, dx_split=False,
                dx_slot=None):
        w_in, bias_in = w, bias
        x, w, bias, residual, gscale = _c(x), _c(w), _c(bias), _c(residual), _c(gscale)
        _chk(x, w, bias, residual, gscale)
        N, C, H, W = x.shape
        K, C2, R, S = w.shape
        if C2 != C:
            raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (C, C2))
        group_n = 1
        if gscale is not None:
            if gscale.dim() != 1 or N % gscale.numel() != 0:
                raise RuntimeError("conv2d: gscale must be 1-D with a length dividing the batch (%d), got %s" % (N, tuple(gscale.shape)))
            group_n = N // gscale.numel()
        OH = (H + 2 * pad - R) // stride + 1
        OW = (W + 2 * pad - S) // stride + 1
        y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32)
        mr_f, mr = _current_min_rows()
        _conv_fwd_launch(mr_f, x, w, bias, residual, y, stride, pad, relu, gscale, group_n, in_split=x_split)
        ctx.save_for_backward(x, w, y if (relu and not grad_premasked) else None, gscale)
        ctx.cfg = (stride, pad, bias is not None, residual is not None, mr, relu, grad_premasked, relu_input, group_n)
        ctx.x_split = bool(x_split)
        ctx.dx_split = bool(dx_split)
        ctx.dx_slot = dx_slot if (dx_split and stride == 1 and C > 4) else None
        ctx.scope = _SCOPE[-1]
        ctx.wkey = _wsink_register(w_in) if ctx.needs_input_grad[1] else None
        ctx.bkey = _wsink_register(bias_in) if (bias_in is not None and ctx.needs_input_grad[2] and _EXTRA_SINKS) else None
        ctx.w_alive = (w_in, bias_in)   # the registry keys on id(): keep the objects alive as long as this node, or a temporary
        return y                        # (a weight slice) could be collected and its id handed to another call's weight

    @staticmethod
    @_scoped_backward
    def backward(ctx, dy):
        x, w, y, gscale = ctx.saved_tensors
        stride, pad, has_bias, has_res, mr, relu, grad_premasked, relu_input, group_n = ctx.cfg
        dy_in = dy
        dy = dy.contiguous()
        N, C, H, W = x.shape
        K, _, R, S = w.shape
        dx = dw = db = dres = dgs = None
        st = _stream()
        dys = is_split(dy_in)       # the gradient arrives as split words (written so by the SPADE backward: ops.conv_split_dy_ok)
        _check_taken_offer(ctx, 0, dy_in, "conv2d")
        if dys and ((relu and not grad_premasked) or (has_res and not getattr(ctx, '_ag2v_res_linked', False)) or dy is not dy_in):
            raise RuntimeError("conv2d backward: a split-word gradient reached a convolution with a ReLU / residual epilogue")
        if relu and not grad_premasked:
            g = torch.empty_like(dy)
            _lib.call('ag2v_act_bwd', 0, _ptr(dy), _ptr(y), _ptr(g), dy.numel(), 0.0, st)   # op 0 = LeakyReLU(slope 0)
            dy = g
        if (ctx.needs_input_grad[0] and _FEWIN_DGRAD and K <= 4 and C >= 16 and stride == 1 and R == S == 3 and pad == 1 and W % 4 == 0
                and tuple(dy.shape[2:]) == (H, W) and not relu_input and gscale is None and not dys and not ctx.dx_split and N <= 65535
                and ctx.dx_slot is None):
            # <= 4 output channels (conv_img, conv_flow): the data gradient is a same-size convolution of the 3- / 2-channel dy to C
            # outputs, 27 / 18 FMAs per written float -- memory-bound on its output.  It runs on the image path of the layout-convolution
            # kernel (4 pixels x 16 channels per thread, filters broadcast from LDS, 16-byte stores; no layout term: J = 0) with the
            # flipped, transposed filter; as a GEMM with K = 27 it had one k-step per tile and ran at 19 / 7 TF.
            dx = torch.empty_like(x)
            wt = w.flip(2, 3).transpose(0, 1).contiguous()
            _lib.call('ag2v_layout_conv_img_fwd', None, None, None, None, _ptr(dx), N, 0, C, H, W, R, 1, pad, 0, _ptr(dy), _ptr(wt), None, K, st)
        elif ctx.needs_input_grad[0]:
            # (the producer of x collects its outputs' gradients in one buffer: this one goes straight into its slice)
            dx = ctx.dx_slot[0].view(ctx.dx_slot[1], N, H, W, x.device) if ctx.dx_slot is not None else torch.empty_like(x)
            if tuple(dx.shape) != tuple(x.shape):
                raise RuntimeError("conv2d backward: gradient slot of shape %s for an input of shape %s" % (tuple(dx.shape), tuple(x.shape)))
            # relu_input: x = relu(a convolution's output) -- that convolution's bias gradient is the channel sum of this dx: the
            # launch adds it up in its epilogue where it can (_CHAN_SUMS_IN_DGRAD; else the producer's backward sums dx as before)
            csum = None
            if relu_input and _CHAN_SUMS_IN_DGRAD and not _DETERMINISTIC[0]:
                csum = ctx.dx_slot[0].sums_view(ctx.dx_slot[1]) if ctx.dx_slot is not None else torch.zeros(C, device=x.device, dtype=torch.float32)
            done = _conv_dgrad_launch(mr, dy, w, x if relu_input else None, dx, stride, pad, 0, gscale, group_n, dy_split=dys,
                                      out_split=ctx.dx_split, chan_sums=csum)
            if ctx.dx_slot is not None:
                ctx.dx_slot[0].mark(ctx.dx_slot[1], summed=bool(done))
            elif done:
                _attach_channel_sums(dx, csum)
            if ctx.dx_split:
                mark_split(dx)
        if gscale is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[4]):
            # per-group weight gradients G_t = dL/d(scale_t W), then dw = sum_t scale_t G_t and dscale_t = <G_t, W> in one pass
            T = gscale.numel()
            G = torch.empty((T,) + tuple(w.shape), device=dy.device, dtype=torch.float32)
            if _wgrad_swap_pays(mr, x.shape, dy.shape, K, R, S, stride, pad, ctx.x_split, dys):
                _conv_wgrad_swapped(mr, dy, x, G, K, R, S, pad, 0, T, ctx.x_split, dys)
            elif _wgrad_pad_pays(mr, x.shape, dy.shape, K, R, S, stride, pad, ctx.x_split, dys):      # (see the plain branch below)
                extra = -dy.shape[3] % 8
                _conv_wgrad_launch(mr, torch.nn.functional.pad(dy, (0, extra)), torch.nn.functional.pad(x, (0, extra)), G, K, R, S, stride, pad, 0,
                                   groups=T)
            else:
                _conv_wgrad_launch(mr, dy, x, G, K, R, S, stride, pad, 0, groups=T, in_split=ctx.x_split, dy_split=dys)
            buf, acc, dw = _wsink_take(ctx.wkey, w) if ctx.needs_input_grad[1] else (torch.empty_like(w), 0, None)
            dgs = torch.empty(T, device=dy.device, dtype=torch.float32)
            _lib.call('ag2v_sn_group_combine', _ptr(G), _ptr(w), _ptr(gscale), T, w.numel(), _ptr(buf), acc | _det_bit(), _ptr(dgs), st)
            if not ctx.needs_input_grad[4]:
                dgs = None
        elif ctx.needs_input_grad[1]:
            buf, acc, dw = _wsink_take(ctx.wkey, w)
            if (_SWAP_FEW_WGRAD and not _FEWROW_WGRAD_DIRECT and K <= 4 and stride == 1 and dy.shape[2] == H and dy.shape[3] == W and C >= 16 and R == S
                    and 2 * pad == R - 1 and not ctx.x_split and not dys):
                # <= 4 output channels (conv_img, the flow / mask heads), same-size convolution: as a GEMM this has 3 rows on a
                # 32-row tile and gathers x once per tap column (0.38 ms at 8 x 64 x 256 x 256).  With the operands swapped --
                # x takes dy's role (C dense rows, read once with 16-byte loads), dy is the gathered 3-channel tensor --
                # the same launch computes dW'[c][m][r'][s'] = dW[m][c][R-1-r'][S-1-s'].
                t = torch.empty((C, K, R, S), device=dy.device, dtype=torch.float32)
                _conv_wgrad_launch(mr, x, dy, t, C, R, S, 1, pad, 0)
                t = t.flip(2, 3).transpose(0, 1)
                if acc:
                    buf.add_(t)
                else:
                    buf.copy_(t)
            elif _wgrad_swap_pays(mr, x.shape, dy.shape, K, R, S, stride, pad, ctx.x_split, dys):
                _conv_wgrad_swapped(mr, dy, x, buf, K, R, S, pad, acc, 1, ctx.x_split, dys)
            elif _wgrad_pad_pays(mr, x.shape, dy.shape, K, R, S, stride, pad, ctx.x_split, dys):
                # stride-1 layers whose output rows are not whole runs of 8 pixels (the discriminators' 4 x 4 layers on 33 x 33 / 17 x 17
                # maps) run their weight gradient on the 4-byte-load kernel (230 TF); with dy and x padded on the right by the same number
                # of zero columns -- the padded dy columns contribute nothing, the padded x columns are the zero padding the real columns
                # saw anyway -- the same sums run on the 16-byte-load kernel
                extra = -dy.shape[3] % 8
                _conv_wgrad_launch(mr, torch.nn.functional.pad(dy, (0, extra)), torch.nn.functional.pad(x, (0, extra)), buf, K, R, S, stride, pad, acc)
            else:
                _conv_wgrad_launch(mr, dy, x, buf, K, R, S, stride, pad, acc, in_split=ctx.x_split, dy_split=dys)
        if has_bias and ctx.needs_input_grad[2]:
            bbuf, bacc, db = _wsink_take(ctx.bkey, ctx.w_alive[1])
            pre = _channel_sums_of(dy_in) if not (relu and not grad_premasked) else None
            if pre is not None and pre.numel() == K and _USE_CHANNEL_SUMS:     # produced with dy by the SPADE backward (_NormAct)
                if bacc:
                    bbuf.add_(pre)
                else:
                    bbuf.copy_(pre)
            else:
                _lib.call('ag2v_channel_sum_acc_s', _ptr(dy), _ptr(bbuf), N, K, dy.shape[2] * dy.shape[3], bacc | _det_bit(), 1 if dys else 0, st)
        if has_res and ctx.needs_input_grad[3]:
            dres = dy_in if dys else dy          # (split words: the very tensor, with its mark, for the residual's producer)
        return dx, dw, db, dres, dgs, None, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, residual=None, relu=False, grad_premasked=False, relu_input=False, gscale=None,
           residual_single_reader=False):
    """y = conv2d(x, w) [* gscale per sample group] + bias + residual   (NCHW, square stride/pad); relu=True: y = max(y, 0).
    grad_premasked / relu_input: see _Conv2d (the pair fuses the ReLU's backward into the consumer's data gradient).
    residual_single_reader: the caller states that nothing else reads `residual`.  When its producer offered to take its gradient as split
    words, y makes the same offer: a split-word d(y) is this convolution's dy AND, unchanged, d(residual)."""
    if relu and residual is not None and grad_premasked:
        raise RuntimeError("conv2d: grad_premasked with a residual input is not supported (the residual's gradient would be masked)")
    res_offer = getattr(residual, '_ag2v_dy_split_ok', None) if (residual is not None and residual_single_reader and _SPLIT_NORM_DX) else None
    if not (isinstance(res_offer, _DySplitOffer) and not res_offer.taken and res_offer.link is None and residual.requires_grad):
        res_offer = None
    dy_ok = (not relu and (residual is None or res_offer is not None) and int(stride) == 1 and torch.is_grad_enabled() and _USE_CHANNEL_SUMS
             and conv_split_dy_ok(x.shape, w.shape, stride, pad))
    # x = relu(...) whose producer takes its gradient as split words (conv2d_shared_input): this convolution's data gradient, which
    # applies that ReLU's mask in its epilogue, writes them
    dx_split = bool(relu_input and dy_ok and x.requires_grad and _take_dy_split(x))
    y = _Conv2d.apply(x, w, bias, residual, gscale, int(stride), int(pad), bool(relu), bool(grad_premasked), bool(relu_input),
                      is_split(x), dx_split, x._ag2v_dy_split_ok.slot if dx_split else None)
    if dy_ok and y.grad_fn is not None:
        _offer_dy_split(y)              # a SPADE backward that is y's only consumer may hand back d(y) as split words
        if residual is not None:
            res_offer.link = y._ag2v_dy_split_ok
            y.grad_fn._ag2v_res_linked = True
    return y


def is_split(t):
    """True when t holds split words (bf16 hi | bf16 lo per value: ag2v_conv_opts::in_split) instead of fp32 -- a tensor written by
    batchnorm_act(..., out_split=True) or a nearest-neighbour resize of one; only bf16x3 convolutions may read it.
    Raises when the tensor was modified in place after it was marked (autograd accumulating a second gradient contribution into it:
    split words added as floats are garbage)."""
    v = getattr(t, '_ag2v_split', None)
    if v is None:
        return False
    if v != t._version:
        raise RuntimeError("ag2video_amd.ops: a split-word tensor was modified in place after it was written (shape %s): it has a second "
                           "consumer or an in-place update, and split words cannot be summed as floats" % (tuple(t.shape),))
    return True


def mark_split(t):
    t._ag2v_split = t._version
    return t


class _DySplitOffer(object):
    """`t._ag2v_dy_split_ok` on an operator output t: the producer's backward accepts d(t) as split words.  The producing autograd
    node keeps the same object (`_ag2v_offers[output_nr]`), so its backward knows whether a consumer took the offer -- and then
    insists on receiving that consumer's split-word tensor (anything else means t had a second consumer)."""
    __slots__ = ('_taken', 'slot', 'link')

    def __init__(self):
        self._taken = False
        self.slot = None        # (_GradSlots, index): where the taker's backward writes d(t) (conv2d_shared_input)
        # t is the residual input of a convolution that passes its own output gradient through unchanged (conv2d(residual=t,
        # residual_single_reader=True)): d(t) is split words exactly when that convolution's offer (this field) was taken
        self.link = None

    @property
    def taken(self):
        return self._taken or (self.link is not None and self.link.taken)

    @taken.setter
    def taken(self, v):
        self._taken = bool(v)


class _GradSlots(object):
    """One [N, sum K_i, H, W] buffer for the gradients of the n outputs of a conv2d_shared_input call.  The call's backward runs ONE
    weight-gradient and ONE data-gradient launch over the concatenated output channels; the consumers of its outputs (one
    conv2d(relu_input=True) each: a SPADE's gamma | beta convolution reading relu(mlp_shared)) write their data gradients straight into
    channel slice i of this buffer instead of n tensors that the backward then copied together (torch.cat: 16 GB read + written per
    c3 step).  Allocated by the first consumer's backward, dropped by the producer's.  A slice counts as the gradient only when a
    consumer's backward of THIS pass has launched its data gradient into it (`written`, set by mark() behind the launch and cleared
    with the buffer): a slice left over from a pass that stopped half way is never read as a gradient."""
    __slots__ = ('ks', 'buf', 'written', 'sums', 'summed', 'handed')

    def __init__(self, ks):
        self.ks = [int(k) for k in ks]
        self.buf = None
        self.written = set()
        self.sums = None          # [sum K_i] channel sums of buf, slice i valid when i is in `summed` (added by the launch that wrote it)
        self.summed = set()
        self.handed = set()

    def view(self, i, N, H, W, device):
        if self.buf is None or tuple(self.buf.shape) != (N, sum(self.ks), H, W):
            self.buf = torch.empty((N, sum(self.ks), H, W), device=device, dtype=torch.float32)
            self.written = set()
            self.sums = None
            self.summed = set()
        self.written.discard(i)
        self.summed.discard(i)
        off = sum(self.ks[:i])
        return self.buf[:, off:off + self.ks[i]]

    def sums_view(self, i):
        """Zeroed slice i of the channel-sum record (call after view(i, ...))."""
        if self.sums is None:
            self.sums = torch.zeros(sum(self.ks), device=self.buf.device, dtype=torch.float32)
            self.handed = set()
        off = sum(self.ks[:i])
        v = self.sums[off:off + self.ks[i]]
        if i in self.handed:         # handed out before in this pass (a retried backward): start from zero again
            v.zero_()
        self.handed.add(i)
        return v

    def mark(self, i, summed=False):
        self.written.add(i)
        if summed:
            self.summed.add(i)

    def all_summed(self):
        return self.sums is not None and all(i in self.summed for i in range(len(self.ks)))

    def drop(self):
        self.buf = None
        self.written = set()
        self.sums = None
        self.summed = set()

    def holds(self, i, t):
        """t is exactly channel slice i of the buffer, and this pass wrote it."""
        b = self.buf
        if b is None or t is None or t.dim() != 4 or i not in self.written:
            return False
        off = sum(self.ks[:i])
        return (t.data_ptr() == b.data_ptr() + off * b.shape[2] * b.shape[3] * 4 and tuple(t.shape) == (b.shape[0], self.ks[i], b.shape[2], b.shape[3])
                and t.stride() == b.stride())


def _offer_dy_split(t):
    offer = _DySplitOffer()
    t._ag2v_dy_split_ok = offer
    node = t.grad_fn
    if node is not None:
        offers = getattr(node, '_ag2v_offers', None)
        if offers is None:
            offers = node._ag2v_offers = {}
        offers[t.output_nr] = offer
    return t


def _take_dy_split(*ts):
    """Consumer side, all or nothing: True when every tensor of ts offers split-word gradients; the offers are then TAKEN (the
    consumer's backward must return split words for each).  A tensor whose offer was taken before has two consumers: raises."""
    offers = [getattr(t, '_ag2v_dy_split_ok', None) for t in ts]
    if not offers or not all(isinstance(o, _DySplitOffer) for o in offers):
        return False
    for o, t in zip(offers, ts):
        if o.taken or o.link is not None:
            raise RuntimeError("ag2video_amd.ops: the gradient of a tensor (shape %s) was promised to one consumer as split words and a "
                               "second consumer reads the same tensor" % (tuple(t.shape),))
    for o in offers:
        o.taken = True
    return True


def _check_taken_offer(ctx, output_nr, dy, what):
    """Producer side (backward): a taken offer's gradient must be the taker's split-word tensor."""
    offers = getattr(ctx, '_ag2v_offers', None)
    o = offers.get(output_nr) if offers else None
    if o is not None and o.taken and (dy is None or not is_split(dy)):
        raise RuntimeError("%s backward: the gradient of output %d was promised as split words and arrived as %s: the output has a "
                           "second consumer (autograd summed split words with floats)" % (what, output_nr, 'None' if dy is None else 'fp32'))


def _attach_channel_sums(g, sums):
    g._ag2v_channel_sums = (sums, g._version)
    return g


def _channel_sums_of(g):
    """The per-channel sums a kernel attached to the gradient tensor g, or None (never attached, or g was modified since)."""
    rec = getattr(g, '_ag2v_channel_sums', None)
    if rec is None or rec[1] != g._version:
        return None
    return rec[0]


def conv_split_dy_ok(x_shape, w_shape, stride, pad):
    """Whether the data gradient and the weight gradient of a convolution of this shape, under the current arithmetic settings, can
    read the OUTPUT gradient as split words (ag2v_conv_opts::dy_split)."""
    N, C, H, W = [int(v) for v in x_shape]
    K, _, R, S = [int(v) for v in w_shape]
    _, mr_b = _current_min_rows()
    return bool(_lib.query('ag2v_conv2d_split_dy_ok', N, C, H, W, K, R, S, int(stride), int(pad), int(mr_b or 0), _PRECISION[0])) \
        and os.environ.get('AG2V_NO_SPLIT_GRAD') is None


def conv_split_input_ok(x_shape, w_shape, stride, pad):
    """Whether a convolution of this shape, called under the CURRENT arithmetic settings (conv precision, bf16x3_min_rows
    context), can read its input as split words in the forward pass and the weight gradient."""
    N, C, H, W = [int(v) for v in x_shape]
    K, _, R, S = [int(v) for v in w_shape]
    mr_f, mr_b = _current_min_rows()
    return bool(_lib.query('ag2v_conv2d_split_input_ok', N, C, H, W, K, R, S, int(stride), int(pad), int(mr_f or 0), int(mr_b or 0),
                           _PRECISION[0]))


class _ConvSharedInput(torch.autograd.Function):
    """y_i = relu?(conv2d(x, w_i) + b_i) for several weight sets on ONE input (the SPADE mlp_shared convolutions of a residual
    block all read the same resized segmentation map).  Same kernels as conv2d; the point is the backward: the input gradient
    is accumulated in place by the data-gradient launches (dx = g_1; dx += g_2; ...) instead of autograd allocating one
    full-size tensor per branch and summing them pairwise."""

    @staticmethod
    def forward(ctx, x, stride, pad, relu, grad_premasked, x_split, out_splits, *wb):
        x = _c(x)
        ctx.wkeys = [(_wsink_register(t) if ctx.needs_input_grad[7 + 2 * i] else None) for i, t in enumerate(wb[0::2])]
        ctx.bkeys = [(_wsink_register(t) if (t is not None and ctx.needs_input_grad[8 + 2 * i] and _EXTRA_SINKS) else None)
                     for i, t in enumerate(wb[1::2])]
        ctx.w_alive = list(wb)           # see _Conv2d.forward
        ws_, bs_ = [_c(t) for t in wb[0::2]], [_c(t) for t in wb[1::2]]
        _chk(x, *ws_)
        _chk(*[b for b in bs_ if b is not None])
        N, C, H, W = x.shape
        mr_f, mr = _current_min_rows()
        outs = []
        if out_splits and not (relu and grad_premasked):
            raise RuntimeError("conv2d_shared_input: split-word outputs need relu=True, grad_premasked=True (the backward never reads them)")
        for w in ws_:
            if w.shape[1] != C:
                raise RuntimeError("conv2d_shared_input: input has %d channels, weight expects %d" % (C, w.shape[1]))
        # ONE launch for the whole group when every branch has 128 output channels, the same filter and the same output format, and the
        # call runs on the LDS-DMA kernels (ag2v_conv_opts::out_mstride): the gathered map -- 16 GB at 256 x 256 -- is then staged once
        # instead of once per branch, and every branch still gets its own contiguous tensor.  Same tiles, same products, same order:
        # bit-identical to the separate launches (tests/test_halo_gpu.py).
        osp = [bool(out_splits and out_splits[j]) for j in range(len(ws_))]
        nb = len(ws_)
        merged = (_MERGE_SHARED_FWD and nb > 1 and all(tuple(w.shape) == tuple(ws_[0].shape) for w in ws_) and ws_[0].shape[0] == 128
                  and len(set(osp)) == 1 and len(set(b is None for b in bs_)) == 1
                  and _fwd_mtile_ok(mr_f, x, 128 * nb, ws_[0].shape[2], ws_[0].shape[3], stride, pad, x_split, osp[0]))
        if merged:
            R, S = ws_[0].shape[2], ws_[0].shape[3]
            OH = (H + 2 * pad - R) // stride + 1
            OW = (W + 2 * pad - S) // stride + 1
            ycat = torch.empty((nb, N, 128, OH, OW), device=x.device, dtype=torch.float32)
            wcat = torch.cat(ws_, dim=0)
            bcat = None if bs_[0] is None else torch.cat(bs_, dim=0)
            _conv_fwd_launch(mr_f, x, wcat, bcat, None, ycat, stride, pad, relu, in_split=x_split, out_split=osp[0],
                             out_mstride=N * 128 * OH * OW)
            outs = list(ycat.unbind(0))
        for j, (w, b) in enumerate(zip(ws_, bs_)):
            if merged:
                break
            K, C2, R, S = w.shape
            OH = (H + 2 * pad - R) // stride + 1
            OW = (W + 2 * pad - S) // stride + 1
            y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.float32)
            _conv_fwd_launch(mr_f, x, w, b, None, y, stride, pad, relu, in_split=x_split, out_split=osp[j])
            outs.append(y)
        keep_y = relu and not grad_premasked
        ctx.save_for_backward(x, *ws_, *(outs if keep_y else []))
        ctx.cfg = (stride, pad, mr, relu, grad_premasked, len(ws_), [b is not None for b in bs_])
        ctx.x_split = bool(x_split)
        ctx.scope = _SCOPE[-1]
        return tuple(outs)

    @staticmethod
    @_scoped_backward
    def backward(ctx, *dys):
        stride, pad, mr, relu, grad_premasked, n, has_b = ctx.cfg
        saved = ctx.saved_tensors
        x, ws_ = saved[0], saved[1:1 + n]
        ys = saved[1 + n:] if (relu and not grad_premasked) else [None] * n
        N, C, H, W = x.shape
        st = _stream()
        dx = None
        grads = []
        live = []                       # (weight, masked dy, index) of the branches that received a gradient
        # gradients written as split words by the consumers' data-gradient epilogues (ops.conv2d: dx_split): all branches or none
        for i_, d_ in enumerate(dys):
            if d_ is not None:
                _check_taken_offer(ctx, i_, d_, "conv2d_shared_input")
        dsp = [is_split(d) for d in dys if d is not None]
        dys_split = bool(dsp) and all(dsp)
        if any(dsp) and (not dys_split or (relu and not grad_premasked)):
            raise RuntimeError("conv2d_shared_input backward: mixed fp32 / split-word gradients")
        # all branches live and all weights want gradients: one weight-gradient launch over the concatenated rows, writing
        # slices of ONE buffer (the per-weight sinks are views of it), so the gathered input is read once per pass
        merged_w = (_MERGE_SHARED_WGRAD and n > 1 and all(d is not None for d in dys)
                    and all(ctx.needs_input_grad[7 + 2 * i] for i in range(n)))
        # the consumers wrote their data gradients into the slices of one buffer (_GradSlots): that buffer IS the concatenation
        slots = getattr(ctx, '_ag2v_slots', None)
        slotted = (slots is not None and merged_w and not (relu and not grad_premasked)
                   and (_MERGE_SHARED_DGRAD or not ctx.needs_input_grad[0]) and all(slots.holds(i, d) for i, d in enumerate(dys)))
        slot_sums = None
        for i in range(n):
            w, dy = ws_[i], dys[i]
            if dy is None:
                grads += [None, None]
                continue
            if not slotted:
                dy = dy.contiguous()
            K, _, R, S = w.shape
            if ys[i] is not None:
                g = torch.empty_like(dy)
                _lib.call('ag2v_act_bwd', 0, _ptr(dy), _ptr(ys[i]), _ptr(g), dy.numel(), 0.0, st)
                dy = g
            live.append((w, dy, i))
            dw = db = None
            if ctx.needs_input_grad[7 + 2 * i] and not merged_w:
                buf, acc, dw = _wsink_take(ctx.wkeys[i], w)
                _conv_wgrad_launch(mr, dy, x, buf, K, R, S, stride, pad, acc, in_split=ctx.x_split, dy_split=dys_split)
            if has_b[i] and ctx.needs_input_grad[8 + 2 * i]:
                bbuf, bacc, db = _wsink_take(ctx.bkeys[i], ctx.w_alive[2 * i + 1])
                if slotted:         # channel sums of the whole buffer, once; each bias takes its slice
                    if slot_sums is None and slots.all_summed():
                        slot_sums = slots.sums          # (every slice's data-gradient launch added its sums in its epilogue)
                    if slot_sums is None:
                        slot_sums = torch.empty(slots.buf.shape[1], device=dy.device, dtype=torch.float32)
                        _lib.call('ag2v_channel_sum_acc_s', _ptr(slots.buf), _ptr(slot_sums), N, slots.buf.shape[1], dy.shape[2] * dy.shape[3],
                                  _det_bit(), 1 if dys_split else 0, st)
                    off = sum(slots.ks[:i])
                    if bacc:
                        bbuf.add_(slot_sums[off:off + K])
                    else:
                        bbuf.copy_(slot_sums[off:off + K])
                else:
                    # (a dense gradient whose producing launch added its channel sums in its epilogue carries them: _attach_channel_sums)
                    pre = _channel_sums_of(dys[i]) if ys[i] is None else None
                    if pre is not None and pre.numel() == K and _USE_CHANNEL_SUMS:
                        if bacc:
                            bbuf.add_(pre)
                        else:
                            bbuf.copy_(pre)
                    else:
                        _lib.call('ag2v_channel_sum_acc_s', _ptr(dy), _ptr(bbuf), N, K, dy.shape[2] * dy.shape[3], bacc | _det_bit(), 1 if dys_split else 0, st)
            grads += [dw, db]
        dyc = None
        if merged_w:
            if slotted:
                dyc = slots.buf
                if dys_split:
                    mark_split(dyc)
            else:
                dyc = torch.cat([d for _, d, _ in live], dim=1)
            Kt = dyc.shape[1]
            R, S = ws_[0].shape[2], ws_[0].shape[3]
            cat, acc, outs = _wsink_take_group(ctx.wkeys, [w for w, _, _ in live])
            _conv_wgrad_launch(mr, dyc, x, cat, Kt, R, S, stride, pad, acc, in_split=ctx.x_split, dy_split=dys_split)
            for j, (_, _, i) in enumerate(live):
                grads[2 * i] = outs[j]
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if not live:
                dx.zero_()
            elif _MERGE_SHARED_DGRAD or len(live) == 1:
                # sum_i dgrad(dy_i, w_i) = dgrad(cat_i dy_i, cat_i w_i): one launch over the concatenated output channels writes
                # dx once (three launches write it three times and read it back twice: 4.3 GB per block at 8 x 512 x 256 x 256,
                # against 1.6 GB for the concatenation) and runs a three times longer reduction per tile
                wc = live[0][0] if len(live) == 1 else torch.cat([w for w, _, _ in live], dim=0)
                if dyc is None:
                    dyc = live[0][1] if len(live) == 1 else torch.cat([d for _, d, _ in live], dim=1)
                _conv_dgrad_launch(mr, dyc, wc, None, dx, stride, pad, 0, dy_split=dys_split)
            else:
                for j, (w, dy, _) in enumerate(live):
                    _conv_dgrad_launch(mr, dy, w, None, dx, stride, pad, 0 if j == 0 else 1, dy_split=dys_split)
        if slots is not None:
            slots.drop()
        return (dx, None, None, None, None, None, None) + tuple(grads)


def conv2d_shared_input(x, weights, biases, stride=1, pad=0, relu=False, grad_premasked=False, out_split=None, grad_split=False):
    """[conv2d(x, w_i, b_i) for i] with one accumulated input gradient (see _ConvSharedInput).  Returns a tuple.
    out_split: per output, store it as split words (relu + grad_premasked outputs whose reader takes them: conv_split_input_ok).
    grad_split: the caller has checked that EVERY consumer (a conv2d(relu_input=True) per output) can write its data gradient as split
    words (conv_split_dy_ok on each of them); the gradients then arrive split when this group's own backward takes them as well."""
    wb = []
    for w, b in zip(weights, biases):
        wb += [w, b]
    flags = tuple(bool(f) for f in out_split) if out_split else None
    if flags is not None and not any(flags):
        flags = None
    outs = _ConvSharedInput.apply(x, int(stride), int(pad), bool(relu), bool(grad_premasked), is_split(x), flags, *wb)
    if flags:
        for o, f in zip(outs, flags):
            if f:
                mark_split(o)
    if (grad_split and relu and grad_premasked and torch.is_grad_enabled()
            and conv_split_dy_ok(x.shape, (sum(int(w.shape[0]) for w in weights),) + tuple(weights[0].shape[1:]), stride, pad)):
        for o in outs:
            if o.grad_fn is not None:
                _offer_dy_split(o)          # the consumers' data gradients may hand d(out) back as split words (all of them or none)
        node = outs[0].grad_fn
        # (small maps -- 8 x 8, 16 x 16: 30 / 120 tiles of 128 x 256 -- keep dense gradients: a data gradient that writes a channel slice
        # cannot split its reduction, and these launches need the splits to fill the chip; the concatenation there is a few MB)
        if (node is not None and len(outs) > 1 and _MERGE_SHARED_WGRAD and _GRAD_SLOTS and int(stride) == 1
                and x.shape[0] * x.shape[2] * x.shape[3] >= _SLOTS_MIN_PIXELS
                and all(o.grad_fn is node for o in outs) and len(set(tuple(w.shape[2:]) for w in weights)) == 1):
            # ... and write them into the slices of one buffer, which this call's backward reads as the concatenation (_GradSlots)
            node._ag2v_slots = _GradSlots([o.shape[1] for o in outs])
            for i, o in enumerate(outs):
                o._ag2v_dy_split_ok.slot = (node._ag2v_slots, i)
    return outs


# ------------------------------------------------------------------------------------------------ spectral norm
class _SpectralWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_orig, u, v, training, eps):
        w = _c(w_orig)
        _chk(w, u, v)
        M = w.shape[0]
        R = w.numel() // M
        wbar = torch.empty_like(w)
        sigma = torch.empty(1, device=w.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_spectral_workspace_floats', M), device=w.device, dtype=torch.float32)
        # the buffers keep advancing in later calls; the tape needs the vectors this call used (written by the same launches)
        need = ctx.needs_input_grad[0]
        uv = torch.empty(M + R, device=w.device, dtype=torch.float32) if need else None
        u_used, v_used = (uv[:M], uv[M:]) if need else (None, None)
        _lib.call('ag2v_spectral_norm_fwd_save', _ptr(w), _ptr(u), _ptr(v), M, R, float(eps), (1 if training else 0) | _det_bit(),
                  _ptr(wbar), _ptr(sigma), _ptr(u_used), _ptr(v_used), _ptr(ws), _stream())
        ctx.save_for_backward(wbar, u_used, v_used, sigma)
        # weight_orig is used by every frame pass of a step: one gradient buffer per backward pass (see _wsink_*)
        ctx.okey = _wsink_register(w_orig) if (ctx.needs_input_grad[0] and _EXTRA_SINKS) else None
        ctx.o_alive = w_orig
        return wbar

    @staticmethod
    def backward(ctx, dwbar):
        wbar, u, v, sigma = ctx.saved_tensors
        dwbar = dwbar.contiguous()
        M = wbar.shape[0]
        R = wbar.numel() // M
        buf, acc, dw = _wsink_take(ctx.okey, wbar)
        ws = torch.empty(_lib.query('ag2v_spectral_workspace_floats', M), device=wbar.device, dtype=torch.float32)
        _lib.call('ag2v_spectral_norm_bwd_acc', _ptr(dwbar), _ptr(wbar), _ptr(u), _ptr(v), _ptr(sigma), M, R, _ptr(buf), acc,
                  _ptr(ws), _stream())
        return dw, None, None, None, None


def spectral_weight(w_orig, u, v, training=True, eps=1e-12):
    """W / sigma(W) with one in-place power iteration on (u, v) in training mode."""
    return _SpectralWeight.apply(w_orig, u, v, bool(training), float(eps))


class _SpectralScales(torch.autograd.Function):
    """scale[t] = 1 / sigma_t for the T train-mode forward calls one spectral-norm convolution makes per step (one per generated
    frame): T consecutive in-place power iterations on (u, v).  conv2d(x, w_orig, gscale=scale) then evaluates all T frame groups
    in one launch.  Eval mode: T = 1 (u, v do not move)."""

    @staticmethod
    def forward(ctx, w_orig, u, v, T, training, eps):
        w = _c(w_orig)
        _chk(w, u, v)
        M = w.shape[0]
        R = w.numel() // M
        if not training:
            T = 1
        sig = torch.empty(2 * T, device=w.device, dtype=torch.float32)
        sigma, scale = sig[:T], sig[T:]
        need = ctx.needs_input_grad[0]
        uv = torch.empty(T * (M + R), device=w.device, dtype=torch.float32) if need else None
        u_used, v_used = (uv[:T * M], uv[T * M:]) if need else (None, None)
        ws = torch.empty(_lib.query('ag2v_spectral_workspace_floats', M), device=w.device, dtype=torch.float32)
        _lib.call('ag2v_spectral_sigmas', _ptr(w), _ptr(u), _ptr(v), M, R, float(eps), (1 if training else 0) | _det_bit(), T, _ptr(sigma),
                  _ptr(scale), _ptr(u_used), _ptr(v_used), _ptr(ws), _stream())
        ctx.save_for_backward(u_used, v_used, scale)
        ctx.cfg = (T, M, R)
        ctx.okey = _wsink_register(w_orig) if (need and _EXTRA_SINKS) else None
        ctx.o_alive = w_orig
        return scale

    @staticmethod
    def backward(ctx, dscale):
        u_used, v_used, scale = ctx.saved_tensors
        T, M, R = ctx.cfg
        dscale = dscale.contiguous()
        buf, acc, dw = _wsink_take(ctx.okey, ctx.o_alive)
        _lib.call('ag2v_spectral_sigmas_bwd', _ptr(dscale), _ptr(scale), _ptr(u_used), _ptr(v_used), T, M, R, _ptr(buf), acc, _stream())
        return dw, None, None, None, None, None


def spectral_scales(w_orig, u, v, T, training=True, eps=1e-12):
    """[1 / sigma_t for t < T] with T in-place power iterations on (u, v) (training) -- or [1 / sigma] in eval mode."""
    return _SpectralScales.apply(w_orig, u, v, int(T), bool(training), float(eps))


class SpectralBank(object):
    """All spectral-norm layers of one network as ONE power-iteration launch chain (ag2v_spectral_sigmas_multi): the generator has 38
    such layers, each making T forward calls per frame-batched step -- 3 launches per power iteration for all of them instead of 4 per
    layer.  ``scales(T)`` runs the T iterations of every layer (advancing weight_u / weight_v in place, as the T forward calls of the
    reference's frame loop do) and returns one [T] tensor of 1/sigma_t per layer, in module order, connected to weight_orig through
    autograd.  Outputs and tape live in buffers owned by the bank and are overwritten by the next call: a backward that arrives after
    a newer forward raises instead of using the wrong vectors."""

    def __init__(self, layers):
        self.layers = list(layers)                # objects with weight_orig, weight_u, weight_v, eps
        self._key = None
        self.version = 0

    def _build(self, T):
        import numpy as np
        ws = [l.weight_orig for l in self.layers]
        key = (T,) + tuple((l.weight_orig.data_ptr(), l.weight_u.data_ptr(), l.weight_v.data_ptr()) for l in self.layers)
        if key == self._key:
            return
        dev = ws[0].device
        Ms = [w.shape[0] for w in ws]
        Rs = [w.numel() // w.shape[0] for w in ws]
        n = len(ws)
        self.sig = torch.empty((n, 2, T), device=dev, dtype=torch.float32)
        self.tape_u = torch.empty(T * sum(Ms), device=dev, dtype=torch.float32)
        self.tape_v = torch.empty(T * sum(Rs), device=dev, dtype=torch.float32)
        self.vtmp = torch.zeros(sum(Rs), device=dev, dtype=torch.float32)
        self.wv = torch.empty(sum(Ms), device=dev, dtype=torch.float32)
        dt = np.dtype([('w', '<u8'), ('u', '<u8'), ('v', '<u8'), ('vtmp', '<u8'), ('wv', '<u8'), ('sigma', '<u8'), ('inv', '<u8'),
                       ('uu', '<u8'), ('vu', '<u8'), ('M', '<i4'), ('R', '<i4'), ('blk0', '<i4'), ('row0', '<i4')])
        assert dt.itemsize == _lib.query('ag2v_sn_desc_bytes')
        rec = np.zeros(n, dt)
        blk = row = mo = ro = 0
        self.slices = []
        for i, (l, M, R) in enumerate(zip(self.layers, Ms, Rs)):
            rec[i] = (ws[i].data_ptr(), l.weight_u.data_ptr(), l.weight_v.data_ptr(), self.vtmp.data_ptr() + 4 * ro,
                      self.wv.data_ptr() + 4 * mo, self.sig[i, 0].data_ptr(), self.sig[i, 1].data_ptr(),
                      self.tape_u.data_ptr() + 4 * T * mo, self.tape_v.data_ptr() + 4 * T * ro, M, R, blk, row)
            self.slices.append((T * mo, T * M, T * ro, T * R, M, R))
            blk += ((R + 255) // 256) * ((M + 63) // 64)
            row += M
            mo += M
            ro += R
        self.total_blocks, self.total_rows = blk, row
        self.desc = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self._key = key

    def scales(self, T, eps=1e-12):
        for l in self.layers:
            if not l.weight_orig.is_contiguous():
                raise RuntimeError("SpectralBank: weight_orig must be contiguous")
        self._build(int(T))
        return _SpectralScalesMulti.apply(self, int(T), float(eps), *[l.weight_orig for l in self.layers])


class _SpectralScalesMulti(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bank, T, eps, *w_origs):
        _chk(*w_origs)
        _lib.call('ag2v_spectral_sigmas_multi', _ptr(bank.desc), len(w_origs), bank.total_blocks, bank.total_rows, T, eps,
                  1 if _DETERMINISTIC[0] else 0, _stream())
        bank.version += 1
        ctx.bank, ctx.version, ctx.T = bank, bank.version, T
        ctx.okeys = [(_wsink_register(w) if (ctx.needs_input_grad[3 + i] and _EXTRA_SINKS) else None) for i, w in enumerate(w_origs)]
        ctx.o_alive = w_origs
        return tuple(bank.sig[i, 1] for i in range(len(w_origs)))

    @staticmethod
    def backward(ctx, *dscales):
        bank, T = ctx.bank, ctx.T
        if bank.version != ctx.version:
            raise RuntimeError("SpectralBank: the tape of this forward pass was overwritten by a later one (one frame-batched "
                               "forward at a time may be awaiting its backward)")
        grads = []
        st = _stream()
        for i, (w, ds) in enumerate(zip(ctx.o_alive, dscales)):
            if ds is None or not ctx.needs_input_grad[3 + i]:
                grads.append(None)
                continue
            uo, un, vo, vn, M, R = bank.slices[i]
            buf, acc, dw = _wsink_take(ctx.okeys[i], w)
            _lib.call('ag2v_spectral_sigmas_bwd', _ptr(ds.contiguous()), _ptr(bank.sig[i, 1]), _ptr_off(bank.tape_u, uo),
                      _ptr_off(bank.tape_v, vo), T, M, R, _ptr(buf), acc, st)
            grads.append(dw)
        return (None, None, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ batch norm family
def _bn_stats(x, running_mean, running_var, training, momentum, eps, groups=1, partials=None):
    """-> mean, invstd ([groups * C]: sample n uses row n // (N / groups)), N, C, HW, samples per group.
    partials (BNPartials, from _bn_partials_of(x, groups)): the statistics pass over x is replaced by the combination of the partials
    x's producer wrote."""
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    if N % groups != 0:
        raise RuntimeError("batch norm: %d sample groups do not divide the batch of %d" % (groups, N))
    Ng = N // groups
    if training:
        mean = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(groups * C, device=x.device, dtype=torch.float32)
        if partials is not None:
            _lib.call('ag2v_bn_finalize_partials', _ptr(partials.buf), groups, C, partials.nparts, float(eps), float(momentum), _ptr(mean),
                      _ptr(invstd), _ptr(running_mean), _ptr(running_var), _stream())
            return mean, invstd, N, C, HW, Ng
        ws = torch.empty(_lib.query('ag2v_bn_workspace_floats', N, C, HW), device=x.device, dtype=torch.float32)
        _lib.call('ag2v_bn_stats_g', _ptr(x), N, C, HW, Ng, float(eps), float(momentum), _ptr(mean), _ptr(invstd),
                  _ptr(running_mean), _ptr(running_var), _ptr(ws), _stream())
    else:
        mean = running_mean
        invstd = torch.rsqrt(running_var + eps)
        Ng = N                                      # running statistics: one row for every sample
    return mean, invstd, N, C, HW, Ng


class _NormAct(torch.autograd.Function):
    """mode 0: BN(affine w,b) -> LeakyReLU(slope); mode 1: SPADE BN*(1+gamma)+beta -> LeakyReLU(slope); mode 2: as 1 with
    gamma | beta delivered as the two channel halves of ONE tensor p0 = [N, 2C, H, W] (the output of the fused gamma/beta conv).
    groups > 1: batch statistics per group of N / groups consecutive samples (the frame groups of the frame-batched generator:
    the reference normalises every frame pass by itself), running statistics updated once per group, in order."""

    @staticmethod
    def forward(ctx, mode, x, p0, p1, running_mean, running_var, training, momentum, eps, slope, groups, out_split=False,
                dgb_split=False, dx_split=False, partials=None):
        from . import distributed as D
        x, p0, p1 = _c(x), _c(p0), _c(p1)
        _chk(x, p0, p1)
        sync = training and D.sync_bn()
        world = 1
        if sync:
            mean, invstd, N, C, HW, Ng = _bn_stats(x, None, None, True, momentum, eps, groups, partials)
            mean, invstd, world = _sync_combine(mean, invstd, groups, C, Ng * HW, eps, momentum, [(running_mean, running_var)])
        else:
            mean, invstd, N, C, HW, Ng = _bn_stats(x, running_mean, running_var, training, momentum, eps, groups, partials if training else None)
        y = torch.empty_like(x)
        if mode == 0:
            _lib.call('ag2v_norm_apply_gs', 0, _ptr(x), _ptr(mean), _ptr(invstd), _ptr(p0), _ptr(p1), None, None, _ptr(y),
                      N, C, HW, Ng, float(slope), 1 if out_split else 0, _stream())
        else:
            if mode == 2 and tuple(p0.shape) != (N, 2 * C) + tuple(x.shape[2:]):
                raise RuntimeError("spade_norm_fused: gamma|beta tensor must be [N, 2C, H, W]")
            g_ptr, b_ptr = (_ptr(p0), _ptr(p1)) if mode == 1 else (_ptr(p0), _ptr_off(p0, C * HW))
            # timed as an HBM-bound kernel: algorithmic bytes = x, gamma, beta read + y written (4 tensors), recorded in
            # the "flops" slot of the timing record
            _launch('ag2v_norm_apply_gs', 'spade_norm_apply', 4.0 * x.numel() * 4, mode, _ptr(x), _ptr(mean), _ptr(invstd), None, None,
                    g_ptr, b_ptr, _ptr(y), N, C, HW, Ng, float(slope), 1 if out_split else 0, _stream(), shape=(N, C, HW))
        # mode 0 (affine batch norm): the backward recomputes the activation's sign from x, w, b -- y is not kept for it
        sign_from_x = mode == 0
        # lean tape, mode 2: only gamma is read by the backward -- keep a compact copy, let the [N, 2C, H, W] tensor go
        ctx.gcompact = bool(mode == 2 and _LEAN_TAPE[0])
        ctx.gb_shape = tuple(p0.shape) if mode == 2 else None
        keep0 = p0[:, :C].contiguous() if ctx.gcompact else p0
        ctx.save_for_backward(x, y if (slope != 1.0 and not sign_from_x) else None, keep0, mean, invstd, p1 if mode == 0 else None)
        ctx.cfg = (mode, training, slope, N, C, HW, Ng, world if sync else 0)
        ctx.dgb_split = bool(dgb_split) and mode == 2
        ctx.dx_split = bool(dx_split) and mode == 2      # d(x) as split words: x's producer took the offer (spade_norm_fused)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, p0, mean, invstd, bias0 = ctx.saved_tensors
        mode, training, slope, N, C, HW, Ng, sync_world = ctx.cfg
        if not training:
            raise RuntimeError("norm backward is only implemented for batch statistics (training mode)")
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        G = N // Ng
        sums = torch.empty((G, C, 4), device=x.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_bn_workspace_floats', N, C, HW), device=x.device, dtype=torch.float32)
        none = (None,) * 11
        gam = p0 if mode >= 1 else None
        wgt = p0 if mode == 0 else None
        dgamma = dbeta = dgb = None
        if mode == 2:
            dgb = torch.empty(ctx.gb_shape, device=x.device, dtype=torch.float32)
            dg_ptr, db_ptr = _ptr(dgb), _ptr_off(dgb, C * HW)
        elif mode == 1:
            dgamma, dbeta = torch.empty_like(x), torch.empty_like(x)
            dg_ptr, db_ptr = _ptr(dgamma), _ptr(dbeta)
        else:
            dg_ptr = db_ptr = None
        kmode = (3 if ctx.gcompact else 2) if mode == 2 else mode     # 3: gamma is a compact [N, C, H, W] copy (lean tape)
        label = 'spade_norm_bwd' if mode == 2 else 'norm_bwd'
        nbytes = ((7.0 if y is not None or slope == 1.0 else 5.0) if mode == 0 else 11.0) * x.numel() * 4
        args = (kmode, _ptr(dy), _ptr(y), _ptr(x), _ptr(gam), _ptr(wgt), _ptr(bias0), _ptr(mean), _ptr(invstd), _ptr(dx), dg_ptr, db_ptr,
                _ptr(sums), N, C, HW, Ng, float(slope), _ptr(ws), (1 if ctx.dgb_split else 0) | (4 if ctx.dx_split else 0))
        if sync_world:
            # (sum dxhat, sum dxhat * xhat) per (group, channel) over ALL ranks between the two passes (batchnorm.py backward of the
            # reduce / broadcast pair); the affine / bias gradients in `sums` stay per rank (averaged with all the others)
            _lib.call('ag2v_norm_bwd_gs_p', *(args + (1, 0.0, _stream())))
            local = _sync_bwd_sums(sums, 1)
            _launch('ag2v_norm_bwd_gs_p', label, nbytes, *(args + (2, 1.0 / (float(sync_world) * Ng * HW), _stream())), shape=(N, C, HW))
            sums = local
        else:
            _launch('ag2v_norm_bwd_gs', label, nbytes, *(args + (_stream(),)), shape=(N, C, HW))
        if ctx.dx_split:
            mark_split(dx)
        if mode == 0:
            dw = db = None
            if p0 is not None:
                tot = sums.sum(dim=0) if G > 1 else sums[0]
                dw, db = tot[:, 1].contiguous(), tot[:, 0].contiguous()
            return (None, dx, dw, db) + none[:11]
        if mode == 2:
            # the channel sums of dgamma | dbeta came out of the reduce pass: they ARE the bias gradient of the convolution that produced
            # gamma | beta (layout [N, 2C, H, W]: gamma channels first).  Handed to its backward on the gradient tensor itself (sync mode:
            # this rank's own sums -- the bias gradient stays per rank)
            tot = sums.sum(dim=0) if G > 1 else sums[0]
            _attach_channel_sums(dgb, torch.cat([tot[:, 3], tot[:, 2]]))
            if ctx.dgb_split:
                mark_split(dgb)
            return (None, dx, dgb, None) + none[:11]
        return (None, dx, dgamma, dbeta) + none[:11]


def batchnorm_act(x, weight, bias, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1,
                  out_split=False):
    """F.batch_norm (batch statistics + running-stat update when training) followed by LeakyReLU(slope); groups: see _NormAct.
    out_split: the result is stored as split words (is_split) -- for a tensor that only bf16x3 convolutions read, which then skip
    their own fp32 -> (hi, lo) split; the caller checks conv_split_input_ok() for every reader first."""
    y = _NormAct.apply(0, x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps),
                       float(slope), int(groups), bool(out_split), False, False, _bn_partials_of(x, groups) if training else None)
    return mark_split(y) if out_split else y


class _BNActPyramid(torch.autograd.Function):
    """batchnorm_act (affine batch norm + LeakyReLU, optionally split-word output) followed by the nearest-neighbour pyramid of its
    output (nearest_pyramid) as ONE node: the backward forms  g_full + sum of the scattered level gradients  while the batch-norm
    backward loads it (ag2v_norm_bwd_pyr) instead of materialising that sum first (one write + two reads of the full-size tensor).
    Returns one tensor per entry of ``sizes`` (the full size is the normalised tensor itself)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, groups, sizes, out_split, recompute=None,
                partials=None):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        _chk(x, weight, bias)
        world = 0
        if training and _sync_world():
            mean, invstd, N, C, HW, Ng = _bn_stats(x, None, None, True, momentum, eps, groups, partials)
            mean, invstd, world = _sync_combine(mean, invstd, groups, C, Ng * HW, eps, momentum, [(running_mean, running_var)])
        else:
            mean, invstd, N, C, HW, Ng = _bn_stats(x, running_mean, running_var, training, momentum, eps, groups, partials if training else None)
        ctx.sync_world = world
        H, W = x.shape[2], x.shape[3]
        y = torch.empty_like(x)
        built = {(H, W): y}
        if H % 2 == 0 and W % 8 == 0 and (H // 2, W // 2) in sizes:
            # the half-size level comes out of the normalisation pass itself (3.1 ms of resize at c3 -> +0.7 ms of writes)
            yh = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
            _lib.call('ag2v_norm_apply_half', _ptr(x), _ptr(mean), _ptr(invstd), _ptr(weight), _ptr(bias), _ptr(y), _ptr(yh),
                      N, C, H, W, Ng, float(slope), 1 if out_split else 0, _stream())
            built[(H // 2, W // 2)] = yh
        else:
            _lib.call('ag2v_norm_apply_gs', 0, _ptr(x), _ptr(mean), _ptr(invstd), _ptr(weight), _ptr(bias), None, None, _ptr(y),
                      N, C, HW, Ng, float(slope), 1 if out_split else 0, _stream())
        for (h, w) in sorted(set(sizes), key=lambda s_: -s_[0] * s_[1]):       # as _NearestPyramid.forward: level from level
            if (h, w) in built:
                continue
            src = min((s_ for s_ in built if s_[0] % h == 0 and s_[1] % w == 0), key=lambda s_: s_[0] * s_[1])
            t = torch.empty((N, C, h, w), device=x.device, dtype=torch.float32)
            _lib.call('ag2v_nearest_resize_fwd', _ptr(built[src]), _ptr(t), N * C, src[0], src[1], h, w, _stream())
            built[(h, w)] = t
        # lean tape: x is the output of a launch that can be repeated bit for bit (recompute = its operands, layout_conv_img): the
        # backward -- the LAST node of the generator's backward pass, when the rest of the tape is gone -- runs it again instead of
        # keeping [N, C, H, W] floats alive through the whole step
        ctx.n_rc = 0
        if recompute is not None:
            rc_t = [t_ for t_ in recompute if torch.is_tensor(t_) or t_ is None]
            ctx.rc_meta = tuple(v for v in recompute if not (torch.is_tensor(v) or v is None))
            ctx.n_rc = len(rc_t)
            ctx.save_for_backward(None, weight, bias, mean, invstd, *rc_t)
        else:
            ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.cfg = (training, slope, N, C, H, W, Ng, tuple(sizes))
        return tuple(y.view_as(y) if (h, w) == (H, W) else built[(h, w)] for (h, w) in sizes)

    @staticmethod
    def backward(ctx, *grads):
        x, weight, bias, mean, invstd = ctx.saved_tensors[:5]
        training, slope, N, C, H, W, Ng, sizes = ctx.cfg
        if not training:
            raise RuntimeError("norm backward is only implemented for batch statistics (training mode)")
        if ctx.n_rc:
            with torch.no_grad():
                x = _layout_conv_img_run(*(tuple(ctx.saved_tensors[5:5 + ctx.n_rc]) + ctx.rc_meta))
            if tuple(x.shape) != (N, C, H, W):
                raise RuntimeError("batchnorm_act_pyramid: the recomputed input has shape %s, expected %s" % (tuple(x.shape), (N, C, H, W)))
        g_full, lv, hs, ws_, keep = None, [], [], [], []
        for (h, w), g in zip(sizes, grads):
            if g is None:
                continue
            g = g.contiguous()
            keep.append(g)
            if (h, w) == (H, W):
                g_full = g if g_full is None else g_full + g
            else:
                lv.append(g.data_ptr()); hs.append(h); ws_.append(w)
        G = N // Ng
        if len(lv) > 2:
            # Nearest-neighbour grids nest (level h takes source rows i * H / h): the gradients of the small levels are first scattered into
            # the second-largest level (a pass over ~1 GB at c3), so the two full-size passes below gather from two levels instead of
            # five -- each further level was a dependent load behind a branch in both of them (19.2 -> 16 ms at 120 x 512 x 256 x 256)
            order = sorted(range(len(lv)), key=lambda i: -hs[i] * ws_[i])
            tgt = order[1]
            rest = [i for i in order[2:] if hs[tgt] % hs[i] == 0 and ws_[tgt] % ws_[i] == 0]
            if len(rest) == len(order) - 2 and ws_[tgt] % 4 == 0:
                folded = torch.empty((N, C, hs[tgt], ws_[tgt]), device=x.device, dtype=torch.float32)
                fp = (ctypes.c_void_p * len(rest))(*[lv[i] for i in rest])
                fh = (ctypes.c_int * len(rest))(*[hs[i] for i in rest])
                fw = (ctypes.c_int * len(rest))(*[ws_[i] for i in rest])
                _lib.call('ag2v_nearest_pyramid_bwd', lv[tgt], ctypes.cast(fp, ctypes.c_void_p), ctypes.cast(fh, ctypes.c_void_p),
                          ctypes.cast(fw, ctypes.c_void_p), len(rest), _ptr(folded), N * C, hs[tgt], ws_[tgt], _stream())
                keep.append(folded)
                lv, hs, ws_ = [lv[order[0]], folded.data_ptr()], [hs[order[0]], hs[tgt]], [ws_[order[0]], ws_[tgt]]
        dx = torch.empty_like(x)
        sums = torch.empty((G, C, 4), device=x.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_bn_workspace_floats', N, C, H * W), device=x.device, dtype=torch.float32)
        n = len(lv)
        arr_p = (ctypes.c_void_p * max(n, 1))(*lv)
        arr_h = (ctypes.c_int * max(n, 1))(*hs)
        arr_w = (ctypes.c_int * max(n, 1))(*ws_)
        args = (_ptr(g_full), ctypes.cast(arr_p, ctypes.c_void_p), ctypes.cast(arr_h, ctypes.c_void_p), ctypes.cast(arr_w, ctypes.c_void_p), n,
                _ptr(x), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(invstd), _ptr(dx), _ptr(sums), N, C, H, W, Ng, float(slope), _ptr(ws))
        if ctx.sync_world:       # synchronised statistics: the reduce pass, the ranks' sums, the apply pass (see _sync_bwd_sums)
            _lib.call('ag2v_norm_bwd_pyr_p', *(args + (1, 0.0, _stream())))
            local = _sync_bwd_sums(sums, 1)
            _launch('ag2v_norm_bwd_pyr_p', 'norm_bwd', 5.0 * x.numel() * 4, *(args + (2, 1.0 / (float(ctx.sync_world) * Ng * H * W), _stream())),
                    shape=(N, C, H * W))
            sums = local
        else:
            _launch('ag2v_norm_bwd_pyr', 'norm_bwd', 5.0 * x.numel() * 4, *(args + (_stream(),)), shape=(N, C, H * W))
        dw = db = None
        if weight is not None:
            tot = sums.sum(dim=0) if G > 1 else sums[0]
            dw, db = tot[:, 1].contiguous(), tot[:, 0].contiguous()
        return (dx, dw, db) + (None,) * 11


def batchnorm_act_pyramid_supported(x_shape, sizes):
    H, W = int(x_shape[2]), int(x_shape[3])
    return ((W % 4 == 0) and len(sizes) <= 8 and all(H % int(h) == 0 and W % int(w) == 0 for (h, w) in sizes)
            and os.environ.get('AG2V_NO_BN_PYRAMID') is None)


def batchnorm_act_pyramid(x, weight, bias, running_mean, running_var, sizes, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1,
                          out_split=False):
    """nearest_pyramid(batchnorm_act(x, ...), sizes) with a fused backward (see _BNActPyramid); requires
    batchnorm_act_pyramid_supported(x.shape, sizes).  Returns a tuple of tensors (split words when out_split)."""
    sizes = tuple((int(h), int(w)) for (h, w) in sizes)
    rc = getattr(x, '_ag2v_recompute', None) if (_LEAN_TAPE[0] and training) else None
    outs = _BNActPyramid.apply(x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps), float(slope),
                               int(groups), sizes, bool(out_split), rc, _bn_partials_of(x, groups) if training else None)
    if out_split:
        for o in outs:
            mark_split(o)
    return outs


def spade_norm(x, gamma, beta, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1):
    """SPADE modulation: BN_noaffine(x) * (1 + gamma) + beta, then LeakyReLU(slope)."""
    return _NormAct.apply(1, x, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps),
                          float(slope), int(groups))


def spade_norm_fused(x, gb, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1, out_split=False,
                     x_single_reader=False):
    """spade_norm with gamma = gb[:, :C] and beta = gb[:, C:] read in place from the [N, 2C, H, W] output of one convolution
    (and their gradients written in place into one [N, 2C, H, W] tensor).
    x_single_reader: the caller states that nothing else reads x; when x's producer (conv2d) offered to take its output gradient as
    split words, d(x) is written so (its data / weight gradients then skip the fp32 -> (hi, lo) split of dy)."""
    take = bool(training) and gb.requires_grad and _take_dy_split(gb)
    # x from a convolution that reads its output gradient as split words (conv2d's offer) and has no other reader: d(x) is written so
    take_x = bool(training) and bool(x_single_reader) and x.requires_grad and _SPLIT_NORM_DX and _take_dy_split(x)
    y = _NormAct.apply(2, x, gb, None, running_mean, running_var, bool(training), float(momentum), float(eps),
                       float(slope), int(groups), bool(out_split), take, take_x)
    return mark_split(y) if out_split else y


def spade_pair_supported(H, W, up):
    return bool(_lib.query('ag2v_spade_pair_supported', int(H), int(W), int(up)))


class _SpadePair(torch.autograd.Function):
    """(y0, y1) = the two SPADE modulations of ONE input x (batch statistics shared), x read at its resolution before the 2x nearest
    upsampling when up == 2 (see csrc/norm.hip).  gb0 / gb1: [N, 2C, H, W] gamma | beta tensors; training mode only."""

    @staticmethod
    def forward(ctx, xs, gb0, gb1, rm0, rv0, rm1, rv1, up, momentum, eps, slope0, slope1, groups, dgb_split=False, split0=False,
                split1=False, dxs_split=False):
        xs, gb0, gb1 = _c(xs), _c(gb0), _c(gb1)
        _chk(xs, gb0, gb1)
        N, C, Hs, Ws = xs.shape
        H, W = Hs * up, Ws * up
        if tuple(gb0.shape) != (N, 2 * C, H, W) or tuple(gb1.shape) != (N, 2 * C, H, W):
            raise RuntimeError("spade_pair: gamma|beta tensors must be [N, 2C, H, W] = %s" % ((N, 2 * C, H, W),))
        if N % groups != 0:
            raise RuntimeError("spade_pair: %d sample groups do not divide the batch of %d" % (groups, N))
        Ng = N // groups
        st = _stream()
        mean = torch.empty(groups * C, device=xs.device, dtype=torch.float32)
        invstd = torch.empty(groups * C, device=xs.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_bn_workspace_floats', N, C, Hs * Ws), device=xs.device, dtype=torch.float32)
        world = 0
        if _sync_world():        # synchronised statistics: local moments, one all-reduce, both running pairs updated on the host side
            _lib.call('ag2v_spade_pair_stats', _ptr(xs), N, C, Hs * Ws, Ng, up * up, float(eps), float(momentum), _ptr(mean), _ptr(invstd),
                      None, None, None, None, _ptr(ws), st)
            mean, invstd, world = _sync_combine(mean, invstd, groups, C, Ng * H * W, eps, momentum, [(rm0, rv0), (rm1, rv1)])
        else:
            _lib.call('ag2v_spade_pair_stats', _ptr(xs), N, C, Hs * Ws, Ng, up * up, float(eps), float(momentum), _ptr(mean), _ptr(invstd),
                      _ptr(rm0), _ptr(rv0), _ptr(rm1), _ptr(rv1), _ptr(ws), st)
        ctx.sync_world = world
        y0 = torch.empty((N, C, H, W), device=xs.device, dtype=torch.float32)
        y1 = torch.empty_like(y0)
        # algorithmic bytes: x (at its own resolution) + 4 gamma / beta planes read, 2 outputs written
        _launch('ag2v_spade_pair_apply_s', 'spade_pair_apply', (xs.numel() + 6.0 * y0.numel()) * 4, _ptr(xs), _ptr(mean), _ptr(invstd), _ptr(gb0),
                _ptr(gb1), _ptr(y0), _ptr(y1), N, C, H, W, up, Ng, float(slope0), float(slope1), 1 if split0 else 0, 1 if split1 else 0, st,
                shape=(N, C, H * W, up))
        # lean tape: the backward reads gamma_0 / gamma_1 only -- compact copies instead of the two [N, 2C, H, W] tensors
        ctx.gcompact = _LEAN_TAPE[0]
        k0, k1 = (gb0[:, :C].contiguous(), gb1[:, :C].contiguous()) if ctx.gcompact else (gb0, gb1)
        ctx.save_for_backward(xs, k0, k1, y0 if slope0 != 1.0 else None, y1 if slope1 != 1.0 else None, mean, invstd)
        ctx.cfg = (up, slope0, slope1, N, C, H, W, Ng)
        ctx.dgb_split = bool(dgb_split)
        ctx.dxs_split = bool(dxs_split)
        return y0, y1

    @staticmethod
    def backward(ctx, dy0, dy1):
        xs, gb0, gb1, y0, y1, mean, invstd = ctx.saved_tensors
        up, slope0, slope1, N, C, H, W, Ng = ctx.cfg
        z = None
        if dy0 is None or dy1 is None:
            z = torch.zeros((N, C, H, W), device=xs.device, dtype=torch.float32)
        dy0 = z if dy0 is None else dy0.contiguous()
        dy1 = z if dy1 is None else dy1.contiguous()
        G = N // Ng
        dxs = torch.empty_like(xs)
        dgb0 = torch.empty((N, 2 * C, H, W), device=xs.device, dtype=torch.float32)
        dgb1 = torch.empty_like(dgb0)
        sums = torch.empty((G, C, 8), device=xs.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_spade_pair_workspace_floats', N, C, H * W), device=xs.device, dtype=torch.float32)
        # algorithmic bytes: reduce pass reads dy0, dy1, gamma0, gamma1 and x (input size) plus y_i only where slope_i != 1 (the
        # kernels skip the LeakyReLU sign otherwise: norm.hip `if (slope != 1.f)`); the apply pass reads the same and writes
        # dgamma / dbeta of both norms and dx at input size: (12 + 2 ny) output-sized passes + 3 input-sized ones (PMC: profiles/r02/pmc)
        ny = (1 if slope0 != 1.0 else 0) + (1 if slope1 != 1.0 else 0)
        nbytes = (3.0 * xs.numel() + (12.0 + 2.0 * ny) * dy0.numel()) * 4
        args = (_ptr(dy0), _ptr(y0), _ptr(dy1), _ptr(y1), _ptr(xs), _ptr(gb0), _ptr(gb1), _ptr(mean), _ptr(invstd), _ptr(dxs), _ptr(dgb0),
                _ptr(dgb1), _ptr(sums), N, C, H, W, up, Ng, float(slope0), float(slope1), _ptr(ws),
                (1 if ctx.dgb_split else 0) | (2 if ctx.gcompact else 0) | (4 if ctx.dxs_split else 0))
        if ctx.sync_world:       # synchronised statistics: the reduce pass, the ranks' sums, the apply pass (see _sync_bwd_sums)
            _lib.call('ag2v_spade_pair_bwd_sp', *(args + (1, 0.0, _stream())))
            local = _sync_bwd_sums(sums, 2)
            _launch('ag2v_spade_pair_bwd_sp', 'spade_pair_bwd', nbytes, *(args + (2, 1.0 / (float(ctx.sync_world) * Ng * H * W), _stream())),
                    shape=(N, C, H * W, up))
            sums = local
        else:
            _launch('ag2v_spade_pair_bwd_s', 'spade_pair_bwd', nbytes, *(args + (_stream(),)), shape=(N, C, H * W, up))
        tot = sums.sum(dim=0) if G > 1 else sums[0]
        _attach_channel_sums(dgb0, torch.cat([tot[:, 3], tot[:, 2]]))      # bias gradients of the two gamma | beta convolutions
        _attach_channel_sums(dgb1, torch.cat([tot[:, 7], tot[:, 6]]))
        if ctx.dgb_split:
            mark_split(dgb0); mark_split(dgb1)
        if ctx.dxs_split:
            mark_split(dxs)
        return (dxs, dgb0, dgb1) + (None,) * 14


def spade_pair(xs, gb0, gb1, rm0, rv0, rm1, rv1, up=1, momentum=0.1, eps=1e-5, slope0=1.0, slope1=1.0, groups=1, out_split=(False, False),
               xs_single_reader=False):
    """Two SPADE modulations of one input (shared batch statistics; both running-stat pairs updated), the input optionally taken
    before its 2x nearest upsampling (up = 2).  Training mode, batch statistics; requires spade_pair_supported(H, W, up).
    xs_single_reader: nothing else reads xs; when its producer offered (conv2d), d(xs) is written as split words."""
    split = gb0.requires_grad and gb1.requires_grad and _take_dy_split(gb0, gb1)
    take_xs = bool(xs_single_reader) and xs.requires_grad and _SPLIT_NORM_DX and _take_dy_split(xs)
    y0, y1 = _SpadePair.apply(xs, gb0, gb1, rm0, rv0, rm1, rv1, int(up), float(momentum), float(eps), float(slope0), float(slope1),
                              int(groups), split, bool(out_split[0]), bool(out_split[1]), take_xs)
    if out_split[0]:
        mark_split(y0)
    if out_split[1]:
        mark_split(y1)
    return y0, y1


class _InstNormLReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, slope):
        x = _c(x)
        _chk(x)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        y = torch.empty_like(x)
        mean = torch.empty(N * C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(N * C, device=x.device, dtype=torch.float32)
        _lib.call('ag2v_instnorm_lrelu_fwd', _ptr(x), _ptr(y), _ptr(mean), _ptr(invstd), N * C, HW, float(eps), float(slope),
                  _stream())
        ctx.save_for_backward(x, y, mean, invstd)
        ctx.cfg = (slope, N * C, HW)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd = ctx.saved_tensors
        slope, NC, HW = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.call('ag2v_instnorm_lrelu_bwd', _ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(invstd), _ptr(dx), NC, HW,
                  float(slope), _stream())
        return dx, None, None


def instnorm_lrelu(x, eps=1e-5, slope=0.2):
    return _InstNormLReLU.apply(x, float(eps), float(slope))


# ------------------------------------------------------------------------------------------------ layout render / warp
class _LayoutRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vecs, boxes, keep, H, W):
        vecs, boxes = _c(vecs), _c(boxes)
        _chk(vecs, boxes)
        keep = keep.to(torch.int32).contiguous()
        BT, O, D = vecs.shape
        out = torch.empty((BT, D, H, W), device=vecs.device, dtype=torch.float32)
        _lib.call('ag2v_layout_render_fwd', _ptr(vecs), _ptr(boxes), _ptr(keep), _ptr(out), BT, O, D, H, W, _stream())
        ctx.save_for_backward(boxes, keep)
        ctx.cfg = (BT, O, D, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes, keep = ctx.saved_tensors
        BT, O, D, H, W = ctx.cfg
        dout = dout.contiguous()
        dvecs = torch.empty((BT, O, D), device=dout.device, dtype=torch.float32)
        _lib.call('ag2v_layout_render_bwd', _ptr(dout), _ptr(boxes), _ptr(keep), _ptr(dvecs), BT, O, D, H, W, _stream())
        return dvecs, None, None, None, None


def layout_render(vecs, boxes, keep, H, W):
    """vecs[BT,O,D], boxes[BT,O,4] (x0,y0,w,h), keep[BT,O] -> [BT,D,H,W]; gradient flows to vecs only (as in the reference:
    the boxes are ground truth or detached, meta_models.py:53)."""
    return _LayoutRender.apply(vecs, boxes, keep, int(H), int(W))


class _LayoutConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, boxes, keep, residual, H, W, R, stride, pad, pool):
        U, boxes, residual = _c(U), _c(boxes), _c(residual)
        _chk(U, boxes, residual)
        keep = keep.to(torch.int32).contiguous()
        NB, J, K, RS = U.shape
        if RS != R * R:
            raise RuntimeError("layout_conv: U has %d taps, expected %d" % (RS, R * R))
        Hm, Wm = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool else (H, W)
        OH = (Hm + 2 * pad - R) // stride + 1
        OW = (Wm + 2 * pad - R) // stride + 1
        out = torch.empty((NB, K, OH, OW), device=U.device, dtype=torch.float32)
        _lib.call('ag2v_layout_conv_fwd', _ptr(U), _ptr(boxes), _ptr(keep), _ptr(residual), _ptr(out), NB, J, K, H, W, R, stride,
                  pad, pool, _stream())
        ctx.save_for_backward(boxes, keep)
        ctx.cfg = (NB, J, K, H, W, R, stride, pad, pool, residual is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes, keep = ctx.saved_tensors
        NB, J, K, H, W, R, stride, pad, pool, has_res = ctx.cfg
        dout = dout.contiguous()
        dU = None
        if ctx.needs_input_grad[0]:
            dU = torch.empty((NB, J, K, R * R), device=dout.device, dtype=torch.float32)
            _lib.call('ag2v_layout_conv_bwd', _ptr(dout), _ptr(boxes), _ptr(keep), _ptr(dU), NB, J, K, H, W, R, stride, pad, pool,
                      _stream())
        dres = dout if (has_res and ctx.needs_input_grad[3]) else None
        return dU, None, None, dres, None, None, None, None, None, None


def layout_conv(U, boxes, keep, H, W, R=3, stride=1, pad=1, residual=None, pool=0):
    """Convolution of a box layout in its low-rank form (see csrc/layout_warp.hip): U[NB,J,K,R*R] are the filter taps already
    contracted with each object's vector; out = residual + sum_j sum_rs U[:,j,:,rs] * w_j(shifted).  pool=1: the convolution
    runs on the 3x3/stride-2 average-pooled layout (the discriminator's second scale)."""
    return _LayoutConv.apply(U, boxes, keep, residual, int(H), int(W), int(R), int(stride), int(pad), int(pool))


def _layout_conv_img_run(U, boxes, keep, img, w_img, gscale, H, W, R, pad, stat_groups=0):
    """The forward launch of _LayoutConvImg on contiguous operands (keep: int32).  No atomics: a second call with the same operands
    returns the same bits -- which is what the lean tape's recomputation relies on (_BNActPyramid).
    stat_groups > 0: also the batch-norm statistics of the result for `stat_groups` groups of consecutive samples, as partials
    (ag2v_layout_conv_img_fwd_stats); returns (out, BNPartials or None)."""
    NB, J, K, RS = U.shape
    IC = img.shape[1]
    iscale = None
    if gscale is not None:
        group_n = NB // gscale.numel()
        iscale = gscale.repeat_interleave(group_n) if group_n > 1 else gscale
    out = torch.empty((NB, K, H, W), device=U.device, dtype=torch.float32)
    if stat_groups:
        part = None
        Ng = NB // stat_groups if NB % stat_groups == 0 else 0
        if (Ng and _LC_STATS and U.data_ptr() % 16 == 0 and w_img.data_ptr() % 16 == 0
                and _lib.query('ag2v_layout_conv_stats_supported', NB, J, K, H, W, R, 1, pad, 0, IC, Ng)):
            part = torch.empty(_lib.query('ag2v_layout_conv_stats_floats', NB, K, H), device=U.device, dtype=torch.float32)
            _lib.call('ag2v_layout_conv_img_fwd_stats', _ptr(U), _ptr(boxes), _ptr(keep), None, _ptr(out), NB, J, K, H, W, R, 1, pad, 0,
                      _ptr(img), _ptr(w_img), _ptr(iscale), IC, _ptr(part), Ng, _stream())
            return out, BNPartials(part, stat_groups, K, Ng * (-(-H // 4)))
    _lib.call('ag2v_layout_conv_img_fwd', _ptr(U), _ptr(boxes), _ptr(keep), None, _ptr(out), NB, J, K, H, W, R, 1, pad, 0,
              _ptr(img), _ptr(w_img), _ptr(iscale), IC, _stream())
    return (out, None) if stat_groups else out


class BNPartials(object):
    """Batch-norm statistics of a tensor as Chan partials [groups, C, nparts, 3], written by the kernel that produced the tensor; kept on the
    tensor as `_ag2v_bn_partials` together with its version (a tensor modified since is not what the partials describe)."""
    __slots__ = ('buf', 'groups', 'C', 'nparts', 'version')

    def __init__(self, buf, groups, C, nparts):
        self.buf, self.groups, self.C, self.nparts, self.version = buf, int(groups), int(C), int(nparts), None


def _bn_partials_of(x, groups):
    """The partials attached to x when they describe x as it is now, for this grouping; else None."""
    p = getattr(x, '_ag2v_bn_partials', None)
    if isinstance(p, BNPartials) and p.version == x._version and p.groups == int(groups) and p.C == x.shape[1]:
        return p
    return None


class _LayoutConvImg(torch.autograd.Function):
    """layout_conv plus the image channels of the same convolution in ONE forward launch:
    out = layout_conv(U) + gscale[group(n)] * conv(img, w_img)   (stride 1, same size).  The backward of the image part is a plain
    convolution's (few-channel data gradient, per-group weight gradients + ag2v_sn_group_combine when gscale is given)."""

    @staticmethod
    def forward(ctx, U, boxes, keep, img, w_img, gscale, H, W, R, pad):
        w_in = w_img
        U, boxes, img, w_img, gscale = _c(U), _c(boxes), _c(img), _c(w_img), _c(gscale)
        _chk(U, boxes, img, w_img, gscale)
        keep = keep.to(torch.int32).contiguous()
        NB, J, K, RS = U.shape
        IC = img.shape[1]
        if RS != R * R or tuple(w_img.shape) != (K, IC, R, R) or tuple(img.shape) != (NB, IC, H, W):
            raise RuntimeError("layout_conv_img: inconsistent shapes")
        group_n = 1 if gscale is None else NB // gscale.numel()
        # (inside ops.frame_groups(T): the batch norm that follows takes its statistics per frame group -- written by this launch)
        out, part = _layout_conv_img_run(U, boxes, keep, img, w_img, gscale, H, W, R, pad, stat_groups=current_frame_groups())
        _LC_LAST_PARTIALS[0] = part
        ctx.save_for_backward(boxes, keep, img, w_img, gscale)
        ctx.cfg = (NB, J, K, H, W, R, pad, group_n, _current_min_rows()[1])
        ctx.scope = _SCOPE[-1]
        ctx.wkey = _wsink_register(w_in) if ctx.needs_input_grad[4] else None
        ctx.w_alive = w_in
        return out

    @staticmethod
    @_scoped_backward
    def backward(ctx, dout):
        boxes, keep, img, w_img, gscale = ctx.saved_tensors
        NB, J, K, H, W, R, pad, group_n, mr = ctx.cfg
        dout = dout.contiguous()
        st = _stream()
        dU = dimg = dw = dgs = None
        if ctx.needs_input_grad[0]:
            dU = torch.empty((NB, J, K, R * R), device=dout.device, dtype=torch.float32)
            _lib.call('ag2v_layout_conv_bwd', _ptr(dout), _ptr(boxes), _ptr(keep), _ptr(dU), NB, J, K, H, W, R, 1, pad, 0, st)
        if ctx.needs_input_grad[3]:
            dimg = torch.empty_like(img)
            _conv_dgrad_launch(mr, dout, w_img, None, dimg, 1, pad, 0, gscale, group_n)
        if gscale is not None and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]):
            T = gscale.numel()
            G = torch.empty((T,) + tuple(w_img.shape), device=dout.device, dtype=torch.float32)
            if _FEWROW_WGRAD_DIRECT and img.shape[1] <= 4 and K <= 128 and R in (3, 4) and 2 * pad == R - 1:
                # <= 4 image channels and a moderate number of outputs (the flow network's first layer): with the operands swapped the
                # launch has <= 4 rows and runs on the few-row kernel (one workgroup per dout channel and frame group)
                _conv_wgrad_swapped(None, dout, img, G, K, R, R, pad, 0, T, False, False)
            else:
                _conv_wgrad_launch(mr, dout, img, G, K, R, R, 1, pad, 0, groups=T)
            buf, acc, dw = _wsink_take(ctx.wkey, w_img) if ctx.needs_input_grad[4] else (torch.empty_like(w_img), 0, None)
            dgs = torch.empty(T, device=dout.device, dtype=torch.float32)
            _lib.call('ag2v_sn_group_combine', _ptr(G), _ptr(w_img), _ptr(gscale), T, w_img.numel(), _ptr(buf), acc | _det_bit(), _ptr(dgs), st)
            if not ctx.needs_input_grad[5]:
                dgs = None
        elif ctx.needs_input_grad[4]:
            buf, acc, dw = _wsink_take(ctx.wkey, w_img)
            _conv_wgrad_launch(mr, dout, img, buf, K, R, R, 1, pad, acc)
        return dU, None, None, dimg, dw, dgs, None, None, None, None


def layout_conv_img_supported(H, W, R, stride, pool, IC):
    return stride == 1 and pool == 0 and W % 4 == 0 and 0 < IC <= 8 and R in (3, 4)


def layout_conv_img(U, boxes, keep, H, W, R, pad, img, w_img, gscale=None):
    """layout_conv(U, ...) + gscale-scaled conv2d(img, w_img) (stride 1, same-size) in one launch: the layout channels and the image
    channels of conv_dim_in / down_flow[0] (generator.py:66,74,82-83).  2 * pad must equal R - 1."""
    if 2 * pad != R - 1:
        raise RuntimeError("layout_conv_img: same-size convolutions only (2 * pad == R - 1)")
    _LC_LAST_PARTIALS[0] = None
    out = _LayoutConvImg.apply(U, boxes, keep, img, w_img, gscale, int(H), int(W), int(R), int(pad))
    part, _LC_LAST_PARTIALS[0] = _LC_LAST_PARTIALS[0], None
    if part is not None:
        part.version = out._version
        out._ag2v_bn_partials = part
    if _LEAN_TAPE[0]:
        # how to produce this tensor again (detached operands, a few MB): a consumer that would otherwise keep the [N, K, H, W] result for
        # its backward may drop it and re-run the launch instead (batchnorm_act_pyramid)
        with torch.no_grad():
            out._ag2v_recompute = (_c(U.detach()), _c(boxes.detach()), keep.to(torch.int32).contiguous(), _c(img.detach()),
                                   _c(w_img.detach()), None if gscale is None else _c(gscale.detach()), int(H), int(W), int(R), int(pad))
    return out


_LC_STATS = os.environ.get('AG2V_NO_LC_STATS') is None       # layout_conv_img writes the following batch norm's statistics (A/B)
_LC_LAST_PARTIALS = [None]


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow):
        img, flow = _c(img), _c(flow)
        _chk(img, flow)
        N, C, H, W = img.shape
        out = torch.empty_like(img)
        _lib.call('ag2v_flow_warp_fwd', _ptr(img), _ptr(flow), _ptr(out), N, C, H, W, _stream())
        ctx.save_for_backward(img, flow)
        return out

    @staticmethod
    def backward(ctx, dout):
        img, flow = ctx.saved_tensors
        N, C, H, W = img.shape
        dout = dout.contiguous()
        dflow = torch.empty_like(flow)
        dimg = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        _lib.call('ag2v_flow_warp_bwd', _ptr(dout), _ptr(img), _ptr(flow), _ptr(dflow), _ptr(dimg), N, C, H, W, _stream())
        return dimg, dflow


def flow_warp(img, flow):
    return _FlowWarp.apply(img, flow)


class _AvgPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        _chk(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
        _lib.call('ag2v_avgpool3s2_fwd', _ptr(x), _ptr(y), N * C, H, W, _stream())
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        _lib.call('ag2v_avgpool3s2_bwd', _ptr(dy), _ptr(dx), N * C, H, W, _stream())
        return dx


def avgpool3s2(x):
    return _AvgPool3s2.apply(x)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        _chk(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
        _lib.call('ag2v_maxpool2_fwd', _ptr(x), _ptr(y), N * C, H, W, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.call('ag2v_maxpool2_bwd', _ptr(dy), _ptr(x), _ptr(dx), N * C, H, W, _stream())
        return dx


def maxpool2(x):
    """nn.MaxPool2d(kernel_size=2, stride=2) (VGG19.features)."""
    return _MaxPool2.apply(x)


class _NearestResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        x = _c(x)
        _chk(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
        _lib.call('ag2v_nearest_resize_fwd', _ptr(x), _ptr(y), N * C, H, W, OH, OW, _stream())
        ctx.shape = (N, C, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        _lib.call('ag2v_nearest_resize_bwd', _ptr(dy), _ptr(dx), N * C, H, W, OH, OW, _stream())
        return dx, None, None


def nearest_resize(x, OH, OW):
    if x.shape[2] == OH and x.shape[3] == OW:
        return x
    y = _NearestResize.apply(x, int(OH), int(OW))
    return mark_split(y) if is_split(x) else y


class _NearestPyramid(torch.autograd.Function):
    """All nearest-neighbour resizes of one map in one node: forward = one resize per requested size (the full size is the
    map itself); backward = ONE kernel that writes every element of the map's gradient once."""

    @staticmethod
    def forward(ctx, x, sizes):
        x = _c(x)
        _chk(x)
        N, C, H, W = x.shape
        # each level is resized from the smallest ALREADY BUILT level it divides (nearest-neighbour picks src = i * factor, so a chain
        # of integer down-samplings is exact): the half-size level reads the full map, every further one only a quarter of its source,
        # instead of five strided passes over the full-resolution map (12 -> 7 ms per c3 step)
        built = {(H, W): x}
        for (h, w) in sorted(set(sizes), key=lambda s: -s[0] * s[1]):
            if (h, w) in built:
                continue
            src = min((s for s in built if s[0] % h == 0 and s[1] % w == 0), key=lambda s: s[0] * s[1])
            sx = built[src]
            y = torch.empty((N, C, h, w), device=x.device, dtype=torch.float32)
            _lib.call('ag2v_nearest_resize_fwd', _ptr(sx), _ptr(y), N * C, src[0], src[1], h, w, _stream())
            built[(h, w)] = y
        outs = [x.view_as(x) if (h, w) == (H, W) else built[(h, w)] for (h, w) in sizes]
        ctx.cfg = (N, C, H, W, tuple(sizes))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        N, C, H, W, sizes = ctx.cfg
        g_full = None
        lv, hs, ws, keep = [], [], [], []
        for (h, w), g in zip(sizes, grads):
            if g is None:
                continue
            g = g.contiguous()
            keep.append(g)
            if (h, w) == (H, W):
                g_full = g if g_full is None else g_full + g
            else:
                lv.append(g.data_ptr()); hs.append(h); ws.append(w)
        dev = keep[0].device
        dx = torch.empty((N, C, H, W), device=dev, dtype=torch.float32)
        n = len(lv)
        arr_p = (ctypes.c_void_p * max(n, 1))(*lv)
        arr_h = (ctypes.c_int * max(n, 1))(*hs)
        arr_w = (ctypes.c_int * max(n, 1))(*ws)
        _lib.call('ag2v_nearest_pyramid_bwd', _ptr(g_full), ctypes.cast(arr_p, ctypes.c_void_p), ctypes.cast(arr_h, ctypes.c_void_p),
                  ctypes.cast(arr_w, ctypes.c_void_p), n, _ptr(dx), N * C, H, W, _stream())
        return dx, None


def nearest_pyramid(x, sizes):
    """Nearest-neighbour resizes of x (NCHW) to each (h, w) of ``sizes`` -- every one an integer down-sampling of x or x's own
    size, W % 4 == 0 and at most 8 levels -- as one autograd node with a fused backward.  Returns a tuple of tensors."""
    H, W = int(x.shape[2]), int(x.shape[3])
    sizes = tuple((int(h), int(w)) for (h, w) in sizes)
    ok = (W % 4 == 0) and len(sizes) <= 8 and all(H % h == 0 and W % w == 0 for (h, w) in sizes)
    if not ok:
        return tuple(nearest_resize(x, h, w) for (h, w) in sizes)
    outs = _NearestPyramid.apply(x, sizes)
    if is_split(x):       # a nearest-neighbour resize copies words: the levels of a split-word map are split-word maps
        for o in outs:
            mark_split(o)
    return outs


def upsample2x(x):
    return _NearestResize.apply(x, int(x.shape[2] * 2), int(x.shape[3] * 2))


# ------------------------------------------------------------------------------------------------ graph ops
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        w_in = w
        x, w, b = _c(x), _c(w), _c(b)
        _chk(x, w, b)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        _lib.call('ag2v_gemm_f32', _ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, K, 1, 1, K, N, 1 if relu else 0, 0, _stream())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (relu, b is not None)
        # the graph nets run once per frame / time step: one weight-gradient buffer per backward pass (see _wsink_*)
        ctx.lkey = _wsink_register(w_in) if (ctx.needs_input_grad[1] and _EXTRA_SINKS) else None
        ctx.l_alive = w_in
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu, has_b = ctx.cfg
        M, K = x.shape
        N = w.shape[0]
        st = _stream()
        g = dy.contiguous()
        if relu:
            g2 = torch.empty_like(g)
            _lib.call('ag2v_relu_bwd', _ptr(g), _ptr(y), _ptr(g2), g.numel(), st)
            g = g2
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call('ag2v_gemm_f32', _ptr(g), _ptr(w), None, _ptr(dx), M, K, N, N, 1, K, 1, K, 0, 0, st)
        if ctx.needs_input_grad[1]:
            buf, acc, dw = _wsink_take(ctx.lkey, w)
            _lib.call('ag2v_gemm_f32', _ptr(g), _ptr(x), None, _ptr(buf), N, K, M, 1, N, K, 1, K, 0, acc, st)
        if has_b and ctx.needs_input_grad[2]:
            db = torch.empty(N, device=g.device, dtype=torch.float32)
            _lib.call('ag2v_col_sum', _ptr(g), _ptr(db), M, N, st)
        return dx, dw, db, None


def linear(x, w, b=None, relu=False):
    """y = [relu](x @ w.T + b) for x[..., K], w[N, K]."""
    shp = x.shape
    y = _Linear.apply(x.reshape(-1, shp[-1]), w, b, bool(relu))
    return y.reshape(*shp[:-1], w.shape[0])


class _TripleGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, pred, edges):
        obj, pred = _c(obj), _c(pred)
        _chk(obj, pred)
        edges = edges.to(torch.int64).contiguous()
        B, O, Din = obj.shape
        T, Dp = pred.shape[1], pred.shape[2]
        out = torch.empty((B, T, 2 * Din + Dp), device=obj.device, dtype=torch.float32)
        _lib.call('ag2v_triple_gather_fwd', _ptr(obj), _ptr(pred), _ptr(edges), _ptr(out), B, O, T, Din, Dp, _stream())
        ctx.save_for_backward(edges)
        ctx.cfg = (B, O, T, Din, Dp)
        return out

    @staticmethod
    def backward(ctx, dtri):
        (edges,) = ctx.saved_tensors
        B, O, T, Din, Dp = ctx.cfg
        dtri = dtri.contiguous()
        dobj = torch.empty((B, O, Din), device=dtri.device, dtype=torch.float32)
        dpred = torch.empty((B, T, Dp), device=dtri.device, dtype=torch.float32)
        _lib.call('ag2v_triple_gather_bwd', _ptr(dtri), _ptr(edges), _ptr(dobj), _ptr(dpred), B, O, T, Din, Dp, _stream())
        return dobj, dpred, None


def triple_gather(obj, pred, edges):
    """[obj[s] | pred | obj[o]] rows for every edge (graph.py:64-70)."""
    return _TripleGather.apply(obj, pred, edges)


class _TriplePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, newt, edges, valid, O, Hd, Dp):
        newt = _c(newt)
        _chk(newt)
        edges = edges.to(torch.int64).contiguous()
        valid = valid.to(torch.uint8).contiguous()
        B, T = newt.shape[0], newt.shape[1]
        pooled = torch.empty((B, O, Hd), device=newt.device, dtype=torch.float32)
        counts = torch.empty((B, O), device=newt.device, dtype=torch.float32)
        _lib.call('ag2v_triple_pool_fwd', _ptr(newt), _ptr(edges), _ptr(valid), _ptr(pooled), _ptr(counts), B, O, T, Hd, Dp,
                  _stream())
        ctx.save_for_backward(edges, valid, counts)
        ctx.cfg = (B, O, T, Hd, Dp)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        edges, valid, counts = ctx.saved_tensors
        B, O, T, Hd, Dp = ctx.cfg
        dpooled = dpooled.contiguous()
        dnewt = torch.empty((B, T, 2 * Hd + Dp), device=dpooled.device, dtype=torch.float32)
        _lib.call('ag2v_triple_pool_bwd', _ptr(dpooled), None, _ptr(edges), _ptr(valid), _ptr(counts), _ptr(dnewt), B, O, T,
                  Hd, Dp, _stream())
        return dnewt, None, None, None, None, None


def triple_pool(newt, edges, valid, O, Hd, Dp):
    """Masked average pooling of the subject/object halves of new triple vectors onto objects (graph.py:79-100)."""
    return _TriplePool.apply(newt, edges, valid, int(O), int(Hd), int(Dp))


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table):
        table = _c(table)
        _chk(table)
        idx = idx.to(torch.int64).contiguous()
        n = idx.numel()
        D = table.shape[1]
        out = torch.empty((n, D), device=table.device, dtype=torch.float32)
        _lib.call('ag2v_embedding_fwd', _ptr(idx), _ptr(table), _ptr(out), n, D, D, 0, _stream())
        ctx.save_for_backward(idx)
        ctx.cfg = (n, table.shape[0], D, tuple(idx.shape))
        return out.view(*idx.shape, D)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        n, V, D, _ = ctx.cfg
        dout = dout.contiguous()
        dtable = torch.empty((V, D), device=dout.device, dtype=torch.float32)
        _lib.call('ag2v_embedding_bwd', _ptr(idx), _ptr(dout), _ptr(dtable), n, V, D, D, 0, _stream())
        return None, dtable


def embedding(idx, table):
    return _Embedding.apply(idx, table)


# ------------------------------------------------------------------------------------------------ pointwise / losses
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, slope):
        x = _c(x)
        _chk(x)
        y = torch.empty_like(x)
        _lib.call('ag2v_act_fwd', op, _ptr(x), _ptr(y), x.numel(), float(slope), _stream())
        ctx.save_for_backward(y)
        ctx.cfg = (op, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        op, slope = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.call('ag2v_act_bwd', op, _ptr(dy), _ptr(y), _ptr(dx), y.numel(), float(slope), _stream())
        return dx, None, None


def leaky_relu(x, slope=0.2):
    return _Act.apply(x, 0, float(slope))


def tanh(x):
    return _Act.apply(x, 1, 0.0)


class _Reduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, a, b, scale):
        a, b = _c(a), _c(b)
        _chk(a, b)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = torch.empty(_lib.query('ag2v_reduce_workspace_floats'), device=a.device, dtype=torch.float32)
        _lib.call('ag2v_reduce_fwd', op, _ptr(a), _ptr(b), a.numel(), float(scale), _ptr(out), _ptr(ws), _stream())
        ctx.save_for_backward(a, b)
        ctx.cfg = (op, scale)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        op, scale = ctx.cfg
        gout = gout.contiguous().view(1)
        da = torch.empty_like(a)
        _lib.call('ag2v_reduce_bwd', op, _ptr(a), _ptr(b), a.numel(), _ptr(gout), float(scale), _ptr(da), _stream())
        db = None
        if b is not None and ctx.needs_input_grad[2]:
            db = -da
        return None, da, db, None


def l1_mean(a, b):
    """mean |a - b|  (nn.L1Loss)."""
    return _Reduce.apply(0, a, b, 1.0 / a.numel())


def hinge_mean(x, kind):
    """GANLoss hinge (networks/loss.py:57-76): kind 'g': -mean(x); 'd_real': -mean(min(x-1,0)); 'd_fake': -mean(min(-x-1,0))."""
    op = {'g': 1, 'd_real': 2, 'd_fake': 3}[kind]
    return _Reduce.apply(op, x, None, -1.0 / x.numel())
