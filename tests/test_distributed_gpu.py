"""RCCL on a real GPU ("multi-GPU is code, not evidence").

A fresh child process is started with the torchrun launcher (never a re-exec of this process, which has initialised the GPU)
with AG2V_FORCE_DIST=1 and one rank per visible GPU (one on the test box): process-group init with the "nccl" (= RCCL) backend
on the device, broadcast, the in-place bucketed all-reduce over FusedAdam's flat gradient buffers, the collective skip flag,
barrier.  The losses must equal those of the same step run in THIS process without torch.distributed -- to the noise of the
split-K atomics (two runs of the same binary differ in the last bits: DESIGN.md 3.4), i.e. 1e-4 relative, and the all-reduce of
a one-rank world must leave the gradients bit-identical.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_path_on_gpu_matches_single_process(device, tmp_path):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    n = max(1, torch.cuda.device_count())  # one rank per visible GPU (the test box has one); the launcher line is the driver's multi-GPU line
    out = os.path.join(str(tmp_path), 'dist.json')
    env = dict(os.environ, AG2V_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'dist_gpu_child.py'), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.load(open(out))
    assert res['backend'] == 'nccl' and res['world'] == n
    assert res['identity_allreduce'] is (True if n == 1 else None)       # a one-rank mean is the identity, bit for bit
    # per gd_step: the collective NaN-guard flag (one int) + one all-reduce per 128 MB bucket of each flat gradient buffer (the
    # toy's buffers are one bucket each: G, D).  The child builds this step with --sync_bn 0 (per-rank statistics) so that the count
    # holds at every world size: with the default (-1 = the reference's behaviour: synchronised at > 1 rank) every batch-norm
    # forward / backward adds its own all-reduce.
    assert res['allreduce_calls_per_gd_step'] == 3, res['allreduce_calls_per_gd_step']
    assert res['skip_flags'] == [False, True] and res['max_over_ranks'] == float(n)
    assert len(set(res['param_checksums'])) == 1
    e = res['sync_bn_err']                                           # synchronised batch norm vs an fp64 evaluation of the formula:
    assert e[0] <= 1e-5 and e[1] <= 5e-4 and e[2] <= 1e-6 and e[3] <= 1e-5, e          # y, dx, running_mean, running_var
    import math
    assert res['held_step'] and all(math.isfinite(v) for v in res['held_step'].values()), res['held_step']      # 64 CUs held by another stream
    if n == 1:      # a whole --sync_bn 1 step on the fused paths: against the oracle's multi-device formula, and against the unfused paths
        ss = res['sync_step']
        assert ss['n_running_var'] >= 20, ss
        assert ss['loss_vs_oracle'] <= 1e-3 and ss['running_var_vs_oracle'] <= 1e-3, ss
        assert ss['loss_fused_vs_plain'] <= 1e-4 and ss['running_var_fused_vs_plain'] <= 1e-5, ss
        c2 = res['c2_sync_held']      # c2 with RCCL + synchronised BN live, 64 CUs held by another stream through the D phase
        assert c2 and all(math.isfinite(v) for v in c2['losses'].values()), c2
        assert c2['losses_equal'] and c2['params_equal'], c2
        print('c2 step, RCCL world 1, --sync_bn 1: %.1f ms alone, %.1f ms beside 64 held CUs' % (c2['ms_unheld'], c2['ms_held']))
    # the same step without torch.distributed, in this process
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
             '--gconv_hidden_dim', '64', '--gpu_ids', '0']
    opt = make_opt(cater_vocab(), flags)
    ts = TrainStep(opt, device, weights_seed=5)
    clip = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
    _, GL0, _ = ts.gd_step(clip, apply_updates=False)
    _, GL, DL = ts.gd_step(clip)
    _, GG = ts.graph_step(ts.to_device(make_clip_batch(opt.vocab, 2, 5, (64, 64), seed=12)))
    # rank 0's clip is this process's clip; its losses before any update are rank-local at every world size (per-rank batch
    # statistics), the graph step follows an update averaged over the ranks' different clips
    cmp = ((res['G0'], GL0), (res['G'], GL), (res['D'], DL)) + (((res['graph'], GG),) if n == 1 else ())
    for got, ref in cmp:
        for k, v in ref.items():
            assert abs(got[k] - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, got[k], float(v))


def test_sync_bn_two_ranks_distinct_data(device, tmp_path):
    """TWO ranks with DIFFERENT data on the fused synchronised-batch-norm paths (batchnorm_act, batchnorm_act_pyramid,
    spade_norm_fused, spade_pair up = 1 / 2) against an fp64 evaluation of the reference's multi-device formula over the concatenated
    batch (sync_batchnorm/batchnorm.py:80-93,105-145) -- outputs, d(x), d(gamma | beta), per-rank affine gradients, running buffers
    (at world 1 the all-reduce is the identity, so a wrong column set or global count would pass there).  The two ranks
    share this box's one GPU; tests/sync_bn_2rank_child.py says how."""
    out = os.path.join(str(tmp_path), 'sync2.json')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'AG2V_FORCE_DIST'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(HERE, 'sync_bn_2rank_child.py'), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.load(open(out))
    assert res['world'] == 2 and len(res['ranks']) == 2
    print('synchronised batch norm, 2 ranks, rank-distinct data, worst relative errors:', res['ranks'])
    bounds = {'y': 2e-5, 'dx': 5e-4, 'dgb': 1e-4, 'dw': 1e-4, 'db': 1e-4, 'rm': 1e-5, 'rv': 1e-5}
    for rr in res['ranks']:
        assert rr['all_reduce_calls'] >= 10                  # forward [sum, sum of squares] + backward sums, per op
        for op in ('batchnorm_act', 'batchnorm_act_pyramid', 'spade_norm_fused', 'spade_pair_up1', 'spade_pair_up2'):
            for k, v in rr[op].items():
                assert v <= bounds[k], (op, k, v, bounds[k])
