"""Child process of tests/test_distributed_gpu.py: one rank of a torchrun job on a real GPU, RCCL ("nccl") backend.

Launched as ``python -m torch.distributed.run --nproc-per-node N ... tests/dist_gpu_child.py OUT.json`` with
AG2V_FORCE_DIST=1, so that even a world of ONE rank takes every collective code path of the product: process-group init on
the device, parameter/buffer broadcast at TrainStep construction, the in-place bucketed all-reduce over FusedAdam's flat
gradient buffers after each backward, the collective skip flag of the training loop, barrier, max-over-ranks.
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main(out_path):
    from ag2video_amd import distributed as D
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    rank, world, local = D.init_from_env()
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl' and D.is_dist()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
             '--gconv_hidden_dim', '64', '--gpu_ids', str(local)]
    # --sync_bn 0: per-rank statistics at every world size (the default -1 synchronises at > 1 rank and would add one all-reduce per
    # batch-norm call to the count below; the synchronised paths have their own legs further down and in sync_bn_2rank_child.py)
    opt = make_opt(cater_vocab(), flags + ['--sync_bn', '0'])
    torch.manual_seed(100 + rank)                     # rank-dependent default init: the broadcast must make ranks agree
    ts = TrainStep(opt, dev, weights_seed=None if os.environ.get('AG2V_TEST_RANDOM_INIT') else 5)
    calls = {'n': 0, 'elems': 0}
    inner = torch.distributed.all_reduce

    def counted(t, *a, **k):
        calls['n'] += 1
        calls['elems'] += t.numel()
        return inner(t, *a, **k)
    torch.distributed.all_reduce = counted
    clip = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11 + rank))
    # gradients before / after the all-reduce on rank-local data: with world == 1 the mean over ranks is the identity, bit for bit
    _, GL0, DL0 = ts.gd_step(clip, apply_updates=False)
    g_local = ts.optimizer_generator.flat_grad.clone()
    D.average_gradients(ts._gen_params, flat=ts.optimizer_generator.flat_grad)
    same = bool(torch.equal(g_local, ts.optimizer_generator.flat_grad)) if world == 1 else None
    n_before = calls['n']
    _, GL, DL = ts.gd_step(clip)
    n_after = calls['n']
    gclip = ts.to_device(make_clip_batch(opt.vocab, 2, 5, (64, 64), seed=12 + rank))
    _, GG = ts.graph_step(gclip)
    # the same step while ANOTHER stream holds 64 CUs from start to end (what RCCL's kernels or a persistent kernel on the side stream
    # do at > 1 rank): the graph-net programs' grid barriers must complete -- check_health() raises on a timeout flag
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _test_aids import hold_cus
    hold = torch.cuda.Stream()
    with torch.cuda.stream(hold):
        hold_cus(64, 400.0, hold)
    _, GLh, DLh = ts.gd_step(clip, apply_updates=False)
    ts.check_health()
    torch.cuda.synchronize()
    held_step = {k: float(v) for k, v in list(GLh.items()) + list(DLh.items())}
    # synchronised batch norm (--sync_bn 1) through the kernels: local statistics -> all-reduce -> the reference's multi-device
    # formula (sync_batchnorm/batchnorm.py:128-145) -> apply; backward with the all-reduced (sum dxhat, sum dxhat xhat)
    from ag2video_amd import ops
    D.set_sync_bn(True)
    gx = torch.Generator().manual_seed(3)
    xb = (torch.randn(6, 5, 9, 9, generator=gx) * 2.0 + 0.5)
    xb[:, 4] *= 1e-4
    xb = xb.to(dev).requires_grad_(True)
    wb, bb = torch.linspace(0.5, 1.5, 5, device=dev).requires_grad_(True), torch.linspace(-1, 1, 5, device=dev).requires_grad_(True)
    rm, rv = torch.zeros(5, device=dev), torch.ones(5, device=dev)
    yb = ops.batchnorm_act(xb, wb, bb, rm, rv, True, 0.1, 1e-5, 0.2, groups=2)
    wgt = torch.randn(yb.shape, generator=gx).to(dev)
    (yb * wgt).sum().backward()
    xd = xb.detach().double().view(2, 3, 5, 81).requires_grad_(True)
    mu = xd.mean(dim=(1, 3), keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=(1, 3), keepdim=True)
    yr = (xd - mu) * var.clamp(min=1e-5).pow(-0.5) * wb.detach().double().view(1, 1, 5, 1) + bb.detach().double().view(1, 1, 5, 1)
    yr = torch.nn.functional.leaky_relu(yr, 0.2)
    (yr * wgt.double().view(2, 3, 5, 81)).sum().backward()
    cnt = world * 3 * 81                      # every rank holds the same xb: global statistics = local ones, over world x the count
    unb = var.squeeze() * cnt / (cnt - 1)
    rm_ref = 0.1 * (0.9 * mu.squeeze()[0] + mu.squeeze()[1])
    rv_ref = 0.9 * (0.9 * 1.0 + 0.1 * unb[0]) + 0.1 * unb[1]
    sync_bn_err = [float((yb.double().view(2, 3, 5, 81) - yr).abs().max() / yr.abs().max()),
                   float((xb.grad.double().view(2, 3, 5, 81) - xd.grad).abs().max() / xd.grad.abs().max()),
                   float((rm.double() - rm_ref).abs().max()), float(((rv.double() - rv_ref).abs() / rv_ref).max())]
    D.set_sync_bn(False)
    # A whole training step with --sync_bn 1 (the reference's behaviour at > 1 device, sync_batchnorm/batchnorm.py:78-93,105-145) on
    # the product's FAST path -- the batch-norm + pyramid node, the SPADE pair, split-word gradients -- against (a) the oracle with its
    # multi-device formula switched on and (b) the same step with those fusions off.  Every rank holds the same
    # clip here, so global statistics = local ones; the oracle leg runs at world 1 only (its CPU all-reduce has no backend otherwise).
    sync_step = None
    if world == 1:
        from oracle import kernels as K
        from oracle.step import OracleTrainer
        opt_s = make_opt(cater_vocab(), flags + ['--sync_bn', '1'])
        sclip = make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=21)

        def run(env):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                with ops.deterministic():
                    t = TrainStep(opt_s, dev, weights_seed=5)
                    assert D.sync_bn()
                    _, gl, dl = t.gd_step(t.to_device(sclip))
                sd = {k: v.detach().double().cpu() for k, v in t.model.state_dict().items() if k.endswith('running_var')}
                return {k: float(v) for k, v in list(gl.items()) + list(dl.items())}, sd, t
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        fused, rv_f, t_f = run({})
        shapes_G = {k: list(v.shape) for k, v in t_f.model.state_dict().items()}
        shapes_D = {k: list(v.shape) for k, v in t_f.discriminator.img_discriminator.state_dict().items()}
        del t_f
        plain, rv_p, t_p = run({'AG2V_NO_BN_PYRAMID': '1', 'AG2V_NO_SPADE_PAIR': '1', 'AG2V_NO_SPLIT_GRAD': '1'})
        del t_p
        D.set_sync_bn(False)
        K.SYNC_BN = True
        try:
            orc = OracleTrainer(opt_s, shapes_G, shapes_D, 5)
            _, GLo, DLo = orc.gd_step(sclip)
        finally:
            K.SYNC_BN = False
        ref = {k: float(v) for k, v in list(GLo.items()) + list(DLo.items())}
        rv_o = {k: v.detach().double() for k, v in orc.G.items() if k.endswith('running_var')}
        sync_step = {'loss_vs_oracle': max(abs(fused[k] - ref[k]) / max(1.0, abs(ref[k])) for k in ref),
                     'loss_fused_vs_plain': max(abs(fused[k] - plain[k]) / max(1.0, abs(plain[k])) for k in plain),
                     'running_var_vs_oracle': max(float(((rv_f[k] - rv_o[k]).abs() / rv_o[k].abs().clamp(min=1e-6)).max()) for k in rv_o),
                     'running_var_fused_vs_plain': max(float(((rv_f[k] - rv_p[k]).abs() / rv_p[k].abs().clamp(min=1e-6)).max()) for k in rv_p),
                     'n_running_var': len(rv_o)}
    D.set_sync_bn(False)
    # BASELINE.json's c2 (128 x 128, 8 frames, batch 4, default widths) the way an 8-GPU rank runs it: collectives
    # live (RCCL), synchronised batch norm on every fused path, the generator's gradient all-reduce in flight under the D phase -- and
    # a second stream holding 64 CUs from before the step until after its D phase, as RCCL's kernels do at > 1 rank.  Deterministic
    # mode: the held run must equal the unheld one bit for bit (co-residency may delay the graph-net programs, never change them).
    c2_sync_held = None
    if world == 1 and not os.environ.get('AG2V_TEST_SKIP_C2'):
        import bench
        opt2, side2, frames2, batch2 = bench.make_opt('c2')
        opt2.sync_bn = 1
        clip2 = bench.make_clip(opt2, 'c2', batch2, frames2, side2, seed=31)
        runs = []
        for held in (False, True):
            with ops.deterministic():
                t2 = TrainStep(opt2, dev, weights_seed=5)
                assert D.sync_bn()
                b2 = t2.to_device(clip2)
                torch.cuda.synchronize()
                if held:
                    with torch.cuda.stream(hold):
                        hold_cus(64, 450.0, hold)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _, gl2, dl2 = t2.gd_step(b2)
                e1.record()
                t2.check_health()
                torch.cuda.synchronize()
            runs.append(({k: float(v) for k, v in list(gl2.items()) + list(dl2.items())}, e0.elapsed_time(e1),
                         float(sum(p.detach().double().abs().sum() for p in t2.model.parameters()))))
            del t2
        D.set_sync_bn(False)
        c2_sync_held = {'losses': runs[1][0], 'losses_equal': runs[0][0] == runs[1][0], 'params_equal': runs[0][2] == runs[1][2],
                        'ms_unheld': runs[0][1], 'ms_held': runs[1][1]}
    skip = D.any_rank(False, dev), D.any_rank(rank == 0, dev)
    torch.distributed.barrier()
    mx = D.max_over_ranks(float(rank + 1), dev)
    # checksum of all parameters: identical on every rank after identical (averaged) updates
    csum = float(sum(p.detach().double().sum() for p in ts.model.parameters()))
    allsum = torch.tensor([csum], dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(allsum) for _ in range(world)]
    torch.distributed.all_gather(gathered, allsum)
    res = {'rank': rank, 'world': world, 'backend': torch.distributed.get_backend(), 'identity_allreduce': same,
           'allreduce_calls_per_gd_step': n_after - n_before, 'allreduce_elems': calls['elems'],
           'flat_grad_elems': [ts.optimizer_generator.flat_grad.numel(), ts.discriminator.optimizer_d_img.flat_grad.numel(),
                               ts.optimizer_graph.flat_grad.numel()],
           'G': {k: float(v) for k, v in GL.items()}, 'D': {k: float(v) for k, v in DL.items()},
           'G0': {k: float(v) for k, v in GL0.items()}, 'graph': {k: float(v) for k, v in GG.items()},
           'skip_flags': list(skip), 'held_step': held_step, 'c2_sync_held': c2_sync_held, 'sync_bn_err': sync_bn_err, 'sync_step': sync_step, 'max_over_ranks': mx, 'param_checksums': [float(t) for t in gathered]}
    if rank == 0:
        json.dump(res, open(out_path, 'w'))
    D.shutdown()


if __name__ == '__main__':
    main(sys.argv[1])
