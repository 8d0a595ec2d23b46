"""world_size-2 checks of the data-parallel path over gloo on CPU (the GPU path uses the same code over RCCL):
bucketed gradient averaging, parameter broadcast, max-over-ranks timing, and an end-to-end graph step in which two ranks
with different clips end up with identical, correctly averaged updates."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from ag2video_amd import distributed as D
    r, w, _ = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and D.is_dist()
    # --- bucketed averaging with a tiny bucket size (forces several buckets, including one tensor > bucket)
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 7, 1000, 1)]
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    mine = [p.grad.clone() for p in params]
    D.average_gradients(params, bucket_bytes=1024)
    gathered = [[torch.zeros_like(m) for _ in range(world)] for m in mine]
    for m, lst in zip(mine, gathered):
        dist.all_gather(lst, m)
    for p, lst in zip(params, gathered):
        assert torch.allclose(p.grad, sum(lst) / world, atol=1e-6)
    # --- the two-phase form on a flat buffer (FusedAdam's layout): reductions launched, unrelated work, then joined
    flat = torch.randn(3000, generator=g)
    mine_flat = flat.clone()
    pending = D.average_gradients_begin([], bucket_bytes=4096, flat=flat)
    assert len(pending.works) == 3                      # 3000 floats in 1024-float buckets
    other = torch.ones(10).sum()                        # (the discriminator phase stands here)
    pending.finish()
    lst = [torch.zeros_like(mine_flat) for _ in range(world)]
    dist.all_gather(lst, mine_flat)
    assert torch.allclose(flat, sum(lst) / world, atol=1e-6) and float(other) == 10.0 and not pending.works
    os.environ['AG2V_SYNC_ALLREDUCE'] = '1'             # the synchronous fallback: nothing pending
    flat2 = mine_flat.clone()
    p2 = D.average_gradients_begin([], bucket_bytes=4096, flat=flat2)
    assert not p2.works and torch.allclose(flat2, flat, atol=1e-6)
    del os.environ['AG2V_SYNC_ALLREDUCE']
    # --- broadcast
    t = torch.full((4,), float(rank))
    D.broadcast_parameters([t])
    assert float(t.sum()) == 0.0
    assert D.max_over_ranks(float(rank + 1), torch.device('cpu')) == float(world)
    # --- synchronised batch-norm statistics: one all-reduce of [sum, sum of squares]; every rank ends with the statistics of the
    # concatenated batch, the reference's multi-device formula (clamp(biased var, eps)^-1/2, unbiased var for the running buffer)
    gx = torch.Generator().manual_seed(7)
    full = torch.randn(2 * 3, 5, 4, 4, generator=gx) * 2.0 + 0.5                 # [ranks * 3 samples, C, H, W], same on both ranks
    full[:, 4] = 1e-4 * full[:, 4]                                                # a channel whose variance is below eps: the clamp acts
    mine_x = full[rank * 3:(rank + 1) * 3]
    G = 3                                                                         # one sample per group: stats over (rank, h, w)
    xs = mine_x.reshape(G, 1, 5, 16)
    mean_l, var_l = xs.mean(dim=(1, 3)), xs.var(dim=(1, 3), unbiased=False)
    mean, inv_std, unb, n = D.combine_bn_stats(mean_l, var_l, 16, 1e-5)
    ref = torch.stack([full[g::3] for g in range(G)])                             # group g = sample g of every rank
    ref = ref.permute(0, 2, 1, 3, 4).reshape(G, 5, -1).double()
    assert n == 32 and torch.allclose(mean.double(), ref.mean(-1), atol=1e-6)
    assert torch.allclose(unb.double(), ref.var(-1, unbiased=True), rtol=1e-5, atol=1e-9)
    want = ref.var(-1, unbiased=False).clamp(min=1e-5).pow(-0.5)
    assert torch.allclose(inv_std.double(), want, rtol=1e-5) and float(inv_std[:, 4].max()) <= 1e-5 ** -0.5 * (1 + 1e-6)
    # --- end-to-end graph step: different clips per rank, identical parameters afterwards
    from _cpu_ops_shim import oracle_ops
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
             '--gconv_hidden_dim', '64', '--gpu_ids', '-1']
    opt = make_opt(cater_vocab(), flags)
    with oracle_ops():
        torch.manual_seed(rank)                       # different random init per rank: broadcast must fix it
        ts = TrainStep(opt, torch.device('cpu'))
        clip = make_clip_batch(opt.vocab, 2, 4, (64, 64), seed=1234 + rank)
        ts.graph_step(clip, apply_updates=False)
        grads = {n: p.grad.clone() for n, p in ts.model.acts_to_boxes.named_parameters() if p.grad is not None}
        ts.graph_step(clip)
    sd = {n: p.detach().clone() for n, p in ts.model.acts_to_boxes.named_parameters()}
    torch.save({'grads': grads, 'params': sd}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    assert a['grads'].keys() == b['grads'].keys() and len(a['grads']) > 10
    for n in a['grads']:
        assert torch.equal(a['grads'][n], b['grads'][n]), n          # averaged gradients are identical on both ranks
    for n in a['params']:
        assert torch.equal(a['params'][n], b['params'][n]), n        # same start (broadcast) + same update


# --------------------------------------------------------------------------------------------------------------------------------
# engine.gd_step at world 2: the generator's gradient all-reduce in its TWO-PHASE form over a flat gradient
# buffer (launched after G's backward, joined behind the discriminator phase), the discriminator's synchronous one, the collective
# NaN-guard flag, and --sync_bn 1 (batch statistics over the global batch with the reference's multi-device formula).  Each rank
# holds one clip of a two-clip batch; a single process then runs the SAME two clips as one batch with the same formula, and the
# ranks' averaged gradients / losses must equal that run's.  The HIP operators are replaced by the oracle's CPU restatements
# (tests/_cpu_ops_shim.py), the flat-buffer optimizer's update by a test-only CPU step: what is under test is the host
# orchestration and the collectives, which are the same code over RCCL.
def _flat_adam_cpu():
    import math
    from ag2video_amd.optim import FusedAdam

    class FlatAdamCPU(FusedAdam):
        """FusedAdam's flat layout (parameters / gradients / moments as views of four buffers) with the update in torch ops."""

        @torch.no_grad()
        def step(self, closure=None):
            self._adopt_grads()
            g = self.param_groups[0]
            self._step += 1
            b1, b2 = g['betas']
            self.flat_m.mul_(b1).add_(self.flat_grad, alpha=1 - b1)
            self.flat_v.mul_(b2).addcmul_(self.flat_grad, self.flat_grad, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self._step, 1 - b2 ** self._step
            denom = (self.flat_v.sqrt() / math.sqrt(bc2)).add_(g['eps'])
            self.flat_param.addcdiv_(self.flat_m, denom, value=-g['lr'] / bc1)
    return FlatAdamCPU


GD_FLAGS = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '1', '--ngf', '8', '--ndf', '8', '--nff', '8',
            '--gconv_hidden_dim', '64', '--gpu_ids', '-1', '--sync_bn', '1']


def _gd_run(clip, out_path, expect_async):
    """One gd_step without the updates (losses + gradients as the optimizers would see them), then one with them."""
    from _cpu_ops_shim import oracle_ops
    from ag2video_amd import distributed as D
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    from oracle import kernels as K
    opt = make_opt(cater_vocab(), GD_FLAGS)
    K.SYNC_BN = True
    seen = []
    begin = D.average_gradients_begin

    def spy(params, *a, **k):
        p = begin(params, *a, **k)
        seen.append(len(p.works))
        return p
    D.average_gradients_begin = spy
    try:
        with oracle_ops():
            ts = TrainStep(opt, torch.device('cpu'), weights_seed=5, adam_cls=_flat_adam_cpu())
            assert D.sync_bn() == expect_async           # live only when the collectives are
            _, GL, DL = ts.gd_step(clip, apply_updates=False)
            gG, gD = ts.optimizer_generator.flat_grad.clone(), ts.discriminator.optimizer_d_img.flat_grad.clone()
            rm = {n: b.clone() for n, b in ts.model.named_buffers() if n.endswith('running_mean') or n.endswith('running_var')}
            ts.gd_step(clip)
    finally:
        D.average_gradients_begin = begin
        K.SYNC_BN = False
    assert seen and all((n > 0) == expect_async for n in seen), seen      # the two-phase path was (not) taken
    torch.save({'GL': {k: float(v) for k, v in GL.items()}, 'DL': {k: float(v) for k, v in DL.items()}, 'gG': gG, 'gD': gD, 'bufs': rm,
                'pG': ts.optimizer_generator.flat_param.clone(), 'pD': ts.discriminator.optimizer_d_img.flat_param.clone()}, out_path)


def _gd_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from ag2video_amd import distributed as D
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    D.init_from_env(backend='gloo')
    full = make_clip_batch(cater_vocab(), world, 3, (64, 64), seed=77, max_objs=6)
    clip = tuple(t[rank:rank + 1] for t in full[:5]) + (full[5][rank:rank + 1],)
    _gd_run(clip, os.path.join(out_dir, 'gd_rank%d.pt' % rank), expect_async=True)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gd_step_sync_bn(tmp_path):
    world = 2
    mp.spawn(_gd_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    torch.set_num_threads(4)
    full = make_clip_batch(cater_vocab(), world, 3, (64, 64), seed=77, max_objs=6)
    _gd_run(full, os.path.join(str(tmp_path), 'gd_single.pt'), expect_async=False)
    a, b, s = (torch.load(os.path.join(str(tmp_path), n)) for n in ('gd_rank0.pt', 'gd_rank1.pt', 'gd_single.pt'))
    for k in ('gG', 'gD', 'pG', 'pD'):
        assert torch.equal(a[k], b[k]), k                      # averaged gradients and updated parameters: identical on both ranks
    for n in a['bufs']:
        assert torch.equal(a['bufs'][n], b['bufs'][n]), n      # synchronised statistics: identical running buffers
        err = (a['bufs'][n] - s['bufs'][n]).abs().max().item()
        assert err <= 1e-3 * s['bufs'][n].abs().max().item(), (n, err)      # = the one-process statistics of the two-clip batch
    # mean over ranks of the per-rank losses = the loss of the two-clip batch (equal sample counts; the norms see global statistics)
    for key in ('GL', 'DL'):
        for k, v in s[key].items():
            m = 0.5 * (a[key][k] + b[key][k])
            assert abs(m - v) <= 1e-4 * max(1.0, abs(v)), (key, k, m, v)
    # ... and the averaged gradients are the gradients of that batch.  D: element-wise at 1e-4 of the largest entry (measured 3e-5).
    # G: relative L2 <= 1e-2 -- the generator's gradient is ill-conditioned in fp32 (warp derivative, L1 / hinge / LeakyReLU kinks:
    # DESIGN.md 4): the SAME one-process batch with its two samples swapped moves it by 1.5e-3 in this norm, the two-rank run by 2.1e-3.
    err = (a['gD'] - s['gD']).abs().max().item()
    assert err <= 1e-4 * s['gD'].abs().max().item(), ('gD', err, s['gD'].abs().max().item())
    rel = ((a['gG'] - s['gG']).norm() / s['gG'].norm()).item()
    assert rel <= 1e-2, ('gG', rel)
    assert float(a['gG'].abs().sum()) > 0 and float(a['gD'].abs().sum()) > 0


def test_bench_gpus_2_without_two_gpus_fails_loudly():
    """`python bench.py --gpus N` starts N ranks itself.  Where fewer than N GPUs are visible (none here) it must
    refuse with a non-zero exit code and a message -- never run fewer ranks and print a line for them."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    if __import__('torch').cuda.device_count() >= 2:
        pytest.skip('two GPUs visible: the launcher would start them')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'toy', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'refusing to run fewer ranks' in r.stderr, r.stderr[-500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    # a rank started by a launcher whose world disagrees with --gpus refuses as well
    env2 = dict(env, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'toy', '--steps', '1', '--warmup', '0'],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr, r.stderr[-500:]
