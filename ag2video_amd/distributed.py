"""Multi-GPU data parallelism: one process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend on ROCm).

The reference is single-process torch.nn.DataParallel: every forward broadcasts all parameters from GPU 0 (518 MB for G)
and every backward reduce-adds all gradients onto GPU 0.  Here replicas stay resident; the only
data-path collective is one gradient all-reduce per optimizer step, issued over flat buckets sized for the
point-to-point xGMI fabric (7 links x ~153 GB/s per GPU: large buckets amortise the per-collective latency, and
RCCL stripes a large all-reduce over all links).  The step's compute is seconds while 500 MB of gradients take
~1-6 ms, so buckets are reduced right after backward on the compute stream (DESIGN.md "Multi-GPU").
Batch-norm statistics: per process by default (per-GPU batch of 8 clips, the single-device formula); ``--sync_bn 1`` switches to
global-batch statistics with the reference's multi-device formula (set_sync_bn / combine_bn_stats below, sync_batchnorm.py).
``bench.py --gpus N`` starts the N ranks itself (a child ``torch.distributed.run``); under an external torchrun it checks WORLD_SIZE.
"""
import os

import torch
import torch.distributed as dist

BUCKET_BYTES = 128 << 20


def is_dist():
    """True when the collectives must run: world size > 1 (or AG2V_FORCE_DIST=1, used to exercise the RCCL code path on
    a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('AG2V_FORCE_DIST') == '1' 


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('AG2V_FORCE_DIST') == '1' and 'RANK' in os.environ
    if (world > 1 or force) and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _buckets(tensors, bucket_bytes):
    cur, size = [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if cur and size + n > bucket_bytes:
            yield cur
            cur, size = [], 0
        cur.append(t)
        size += n
    if cur:
        yield cur


@torch.no_grad()
def average_gradients(params, bucket_bytes=BUCKET_BYTES, flat=None):
    """grad <- mean over ranks.  ``flat``: the optimizer's flat gradient buffer (ag2video_amd.optim.FusedAdam) -- reduced in
    place in bucket-sized chunks, no copies; otherwise per-parameter gradients are packed into flat buckets.
    No-op on a single process."""
    if not is_dist():
        return
    world = dist.get_world_size()
    if flat is not None:
        step = max(1, bucket_bytes // flat.element_size())
        for off in range(0, flat.numel(), step):
            chunk = flat[off:off + step]
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
            chunk.div_(world)
        return
    grads = [p.grad for p in params if p.grad is not None]
    for bucket in _buckets(grads, bucket_bytes):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


class _Pending(object):
    """Handle of average_gradients_begin: finish() makes the current stream wait for the reductions and turns sums into means."""

    def __init__(self, works, flat, world):
        self.works, self.flat, self.world = works, flat, world

    def finish(self):
        for w in self.works:
            w.wait()            # (NCCL / RCCL: the current stream waits, the host does not)
        if self.flat is not None:
            self.flat.div_(self.world)
        self.works, self.flat = [], None


def average_gradients_begin(params, bucket_bytes=BUCKET_BYTES, flat=None):
    """average_gradients in two halves: the all-reduces of the flat gradient buffer are LAUNCHED here (asynchronously, in the same
    bucket order on every rank) and the caller runs independent work -- the discriminator's loss and backward pass, which read none
    of these gradients -- before it calls .finish() on the returned handle.  Without a flat buffer (torch.optim.Adam on the CPU) or
    with AG2V_SYNC_ALLREDUCE set the reduction happens here and finish() is a no-op."""
    if not is_dist():
        return _Pending([], None, 1)
    if flat is None or os.environ.get('AG2V_SYNC_ALLREDUCE') is not None:
        average_gradients(params, bucket_bytes, flat)
        return _Pending([], None, 1)
    world = dist.get_world_size()
    step = max(1, bucket_bytes // flat.element_size())
    works = [dist.all_reduce(flat[off:off + step], op=dist.ReduceOp.SUM, async_op=True) for off in range(0, flat.numel(), step)]
    return _Pending(works, flat, world)


@torch.no_grad()
def broadcast_parameters(tensors, src=0):
    """Make every rank start from rank 0's parameters and buffers (the reference re-broadcasts every forward)."""
    if not is_dist():
        return
    for t in tensors:
        if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
            dist.broadcast(t, src=src)


def any_rank(flag, device):
    """True on every rank iff ``flag`` is true on at least one rank (one MAX all-reduce of a single int).  The training loop uses
    it to make skip decisions collective: the reference is ONE process, so its NaN guard (scripts/train.py:450-453) and its
    per-step try/except (:466-468) may skip locally; with one process per GPU a rank that skipped alone would leave the other
    ranks inside a gradient all-reduce it never enters."""
    if not is_dist():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


# ---- synchronised batch norm (reference: sync_batchnorm/batchnorm.py:80-93,105-126,128-145).  The reference's DataParallel splits
# the batch over the devices and reduces [sum, sum of squares] per channel onto a master; here every rank all-reduces them.
_SYNC_BN = [False]


def set_sync_bn(flag):
    """Batch statistics over the GLOBAL batch (all ranks) with the reference's multi-device formula -- inv_std =
    clamp(biased var, eps)^-1/2, running_var fed with the unbiased variance -- instead of per-rank statistics with
    F.batch_norm's (var + eps)^-1/2.  Takes effect when the collectives are live (is_dist())."""
    _SYNC_BN[0] = bool(flag)


def sync_bn():
    return _SYNC_BN[0] and is_dist()


@torch.no_grad()
def combine_bn_stats(mean_l, var_l, count_l, eps):
    """Per-rank (mean, biased variance) [G, C] over count_l elements each -> global (mean, inv_std, unbiased var, count) after ONE
    all-reduce of [sum, sum of squares] (batchnorm.py:105-126 reduces the same pair; _compute_mean_std :128-145).  fp64 on the
    wire: E[x^2] - mean^2 cancels badly in fp32."""
    m, v = mean_l.double(), var_l.double()
    pack = torch.stack([m * count_l, (v + m * m) * count_l])
    dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    n = float(count_l) * dist.get_world_size()
    mean = pack[0] / n
    sumvar = pack[1] - pack[0] * mean
    bias_var, unbias_var = sumvar / n, sumvar / (n - 1)
    inv_std = bias_var.clamp(min=eps).pow(-0.5)
    return mean.float(), inv_std.float(), unbias_var.float(), n


@torch.no_grad()
def sum_over_ranks_(t):
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(value, device):
    if not is_dist():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
