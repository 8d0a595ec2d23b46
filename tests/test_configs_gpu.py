"""BASELINE.json's configurations through the HIP path (VERDICT r01 "configs never run through the HIP path": c1, c2, c5).

* c1 (64x64, 4 frames, batch 1, DEFAULT widths: ngf 64 / ndf 64 / nff 32 / hidden 512 -> the 1024-channel layers, spectral norm on
  1024 x 9216 matrices, K = 9*1024 reductions): one full iteration (G step, D step, graph step) on the GPU against the CPU
  oracle on the same seeded clip and closed-form weights -- losses and output tensors within 1e-3 relative, gradient statistics
  at the conditioning bounds of tests/test_step_gpu.py, every post-step state tensor (parameters after Adam, spectral-norm
  u / v, running statistics).  Both arithmetic modes.
* c2 (128x128, 8 frames, batch 4) and a c5-shaped clip (Something-Else vocabulary: one attribute => 128 layout channels, 2-3
  objects, 224x224 = 7 * 32: the non-power-of-two pyramid 7, 14, ..., 224, 16 frames): the oracle cannot run them in seconds,
  so they go through the size-independent property set of tests/test_fullsize_gpu.py at THEIR layer sizes (oracle on crops,
  adjoint identities) plus one finite training step with the documented keys.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def _rand(*shape, seed=0, device='cpu', scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c1_default_widths_matches_oracle(prec, device):
    import bench
    from ag2video_amd import ops
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    from oracle.step import OracleTrainer
    torch.set_num_threads(min(16, torch.get_num_threads()))
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        opt, side, frames, batch = bench.make_opt('c1')
        assert (opt.ngf, opt.ndf, opt.nff, opt.gconv_hidden_dim) == (64, 64, 32, 512)
        opt.gpu_ids = [0]
        ts = TrainStep(opt, device, weights_seed=1)
        shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
        shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
        orc = OracleTrainer(opt, shapes_G, shapes_D, 1)
        clip = make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234)
        mo_r, GL_r, DL_r = orc.gd_step(clip)
        mo, GL, DL = ts.gd_step(ts.to_device(clip))
        for got, ref in ((GL, GL_r), (DL, DL_r)):
            assert set(got) == set(ref)
            for k in ref:
                assert abs(got[k].item() - ref[k].item()) <= 1e-3 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
        worst = {}
        for i, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred')):
            ref = mo_r[i].detach().numpy()
            err = np.abs(mo[i].detach().cpu().numpy() - ref).max()
            worst[name] = err / np.abs(ref).max()
            assert err <= 1e-3 * np.abs(ref).max(), (name, err, np.abs(ref).max())
        assert np.mean(mo[3].cpu().numpy() != mo_r[3].numpy()) < 2e-3
        # gradients: abs-sum of every parameter's gradient (bounds: module docstring of tests/test_step_gpu.py)
        bad = []
        for n, prm in ts.model.named_parameters():
            g_ref = orc.last_G_grads.get(n)
            if g_ref is None:
                assert prm.grad is None or float(prm.grad.abs().sum()) == 0.0 or n.startswith('acts_to_boxes'), n
                continue
            assert prm.grad is not None, n
            got, ref = _stats(prm.grad), _stats(g_ref)
            rel = 5e-2 if ('flows_network' in n or n.startswith('acts_to_objs')) else 2e-2
            if abs(got[1] - ref[1]) > rel * abs(ref[1]) + 2e-6 * prm.numel():
                bad.append((n, got[1], ref[1]))
        assert not bad, (len(bad), bad[:10])
        badd = []
        for n, prm in ts.discriminator.img_discriminator.named_parameters():
            g_ref = orc.last_D_grads.get(n)
            if g_ref is None or prm.grad is None:
                continue
            got, ref = _stats(prm.grad), _stats(g_ref)
            if abs(got[1] - ref[1]) > 2e-2 * abs(ref[1]) + 2e-6 * prm.numel():
                badd.append((n, got[1], ref[1]))
        assert not badd, (len(badd), badd[:10])
        # graph step on a 4x longer clip (train.py:471-493)
        gclip = make_clip_batch(opt.vocab, batch, 4 * frames, (side, side), seed=4321)
        bp_r, GG_r = orc.graph_step(gclip)
        bp, GG = ts.graph_step(ts.to_device(gclip))
        for k in GG_r:
            assert abs(GG[k].item() - GG_r[k].item()) <= 1e-3 * max(1.0, abs(GG_r[k].item())), (k, GG[k].item(), GG_r[k].item())
        assert np.abs(bp.detach().cpu().numpy() - bp_r.detach().numpy()).max() <= 1e-3 * np.abs(bp_r.detach().numpy()).max()
        # every state tensor after the three optimizer steps: buffers (spectral-norm u / v, running statistics) element-wise at
        # 1e-3 of their max; parameters through (abs-sum, L2) at 1e-3 + Adam's first-step floor lr per element (a parameter whose
        # true gradient is zero moves by +-lr in a rounding-determined direction)
        floor = 1.2 * opt.learning_rate
        bad_state, n_keys = [], 0
        for sd, ref_sd in ((ts.model.state_dict(), orc.G), (ts.discriminator.img_discriminator.state_dict(), orc.D)):
            assert set(sd) == set(ref_sd)
            for k, v in sd.items():
                r = ref_sd[k].detach()
                n_keys += 1
                if not ref_sd[k].requires_grad and r.is_floating_point():
                    err = (v.detach().cpu() - r).abs().max().item()
                    if err > 1e-3 * max(r.abs().max().item(), 1e-6):
                        bad_state.append((k, 'buffer', err, r.abs().max().item()))
                got, ref = _stats(v.float()), _stats(r.float())
                n = v.numel()
                for i, fl in ((1, floor * n), (2, floor * n ** 0.5)):
                    if abs(got[i] - ref[i]) > 1e-3 * abs(ref[i]) + fl:
                        bad_state.append((k, i, got[i], ref[i]))
        assert n_keys == 492 + 56
        assert not bad_state, (len(bad_state), bad_state[:10])
        print('c1 default widths [%s]: worst output errors relative to max:' % prec, worst)
    finally:
        ops.set_conv_precision(old)


# ---------------------------------------------------------------------------------------------------- c2 / c5 layer sizes
# (N, C, H, W, K, R, stride, pad): the GEMM shapes these configurations add over c3's (tests/test_fullsize_gpu.py):
LAYERS = {
    'c2_mlp_shared_128': (4, 512, 128, 128, 128, 3, 1, 1),
    'c2_up3_conv0_128': (4, 128, 128, 128, 64, 3, 1, 1),
    'c2_D_model1_65': (28, 64, 65, 65, 128, 4, 2, 2),          # discriminator, 4 x 7 frames, 4x4 stride 2
    'c5_mlp_shared_224': (2, 128, 224, 224, 128, 3, 1, 1),     # Something-Else: 128 layout channels
    'c5_gamma_beta_112': (2, 128, 112, 112, 256, 3, 1, 1),
    'c5_head_7': (2, 1024, 7, 7, 1024, 3, 1, 1),               # 7 x 7 latent map: 49 pixels per sample
    'c5_up0_28': (2, 1024, 28, 28, 512, 3, 1, 1),
    'c5_D_model0_img_224': (30, 3, 224, 224, 64, 4, 2, 2),     # image part of D's first convolution -> 113 x 113
    'c5_D_model2_57': (30, 128, 57, 57, 256, 4, 2, 2),
}


@pytest.mark.parametrize("name", sorted(LAYERS))
@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_conv_windows_and_adjoints_at_config_sizes(name, prec, device):
    """(a) windows of the full-size output (corners included: padding taps, odd sizes) against the oracle on the matching crop;
    (b) <conv(x, w), dy> == <x, dgrad(dy)> == <w, wgrad(dy)> in fp64 over the fp32 kernel outputs."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Ko, R, st, pad = LAYERS[name]
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        with ops.bf16x3_min_rows(48):
            x = _rand(N, C, H, W, seed=1, device=device).requires_grad_(True)
            w = _rand(Ko, C, R, R, seed=2, device=device, scale=(C * R * R) ** -0.5).requires_grad_(True)
            b = _rand(Ko, seed=3, device=device)
            y = ops.conv2d(x, w, b, st, pad)
            OH, OW = y.shape[2], y.shape[3]
            dy = _rand(N, Ko, OH, OW, seed=4, device=device)
            y.backward(dy)
        win = min(12, OH)
        for (n, oy, ox) in [(0, 0, 0), (N - 1, OH - win, OW - win), (N // 2, (OH - win) // 2, (OW - win) // 3)]:
            # input rows/cols feeding output window [oy, oy+win) x [ox, ox+win): [o*st - pad, (o+win-1)*st - pad + R)
            ys, xs = oy * st - pad, ox * st - pad
            ye, xe = (oy + win - 1) * st - pad + R, (ox + win - 1) * st - pad + R
            crop = x.detach()[n:n + 1, :, max(ys, 0):min(ye, H), max(xs, 0):min(xe, W)].cpu()
            crop = torch.nn.functional.pad(crop, (max(-xs, 0), max(xe - W, 0), max(-ys, 0), max(ye - H, 0)))
            ref = K.conv2d(crop, w.detach().cpu(), b.cpu(), st, 0)
            got = y.detach()[n:n + 1, :, oy:oy + win, ox:ox + win].cpu()
            assert ref.shape == got.shape, (ref.shape, got.shape)
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-4, (name, prec, (n, oy, ox), err)
        yb = (y.detach() - b.view(1, -1, 1, 1)).double()
        a = (yb * dy.double()).sum().item()
        bx = (x.detach().double() * x.grad.double()).sum().item()
        bw = (w.detach().double() * w.grad.double()).sum().item()
        scale = (yb.pow(2).sum().sqrt() * dy.double().pow(2).sum().sqrt()).item()
        assert abs(a - bx) <= 2e-5 * scale, (name, a, bx, scale)
        assert abs(a - bw) <= 2e-5 * scale, (name, a, bw, scale)
    finally:
        ops.set_conv_precision(old)


def test_spade_norm_and_pyramid_at_224(device):
    """SPADE modulation moments and the segmentation-map pyramid (7, 14, ..., 224: factor-32 pyramid of a non-power-of-two side)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H = 2, 64, 224
    x = _rand(N, C, H, H, seed=4, device=device) * 3.0 + 1.5
    gb = torch.zeros(N, 2 * C, H, H, device=device)
    rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
    y = ops.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, 1.0)
    m = y.double().mean(dim=(0, 2, 3))
    v = y.double().var(dim=(0, 2, 3), unbiased=False)
    assert m.abs().max().item() <= 1e-4 and (v - 1).abs().max().item() <= 1e-4
    seg = _rand(1, 16, H, H, seed=5)
    sizes = [(7 << i, 7 << i) for i in range(6)]
    sg = seg.clone().to(device).requires_grad_(True)
    sc = seg.clone().requires_grad_(True)
    got = ops.nearest_pyramid(sg, sizes)
    lg = lr = 0.0
    for i, ((h, w), yg) in enumerate(zip(sizes, got)):
        yr = K.nearest_resize(sc, h, w)
        assert torch.equal(yg.detach().cpu(), yr.detach()), (h, w)
        wg = _rand(*yr.shape, seed=20 + i)
        lg = lg + (yg * wg.to(device)).sum()
        lr = lr + (yr * wg).sum()
    lg.backward()
    lr.backward()
    assert (sg.grad.cpu() - sc.grad).abs().max().item() <= 1e-5 * sc.grad.abs().max().item()


def _finite_step(cfg, device, batch=None):
    import bench
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, b = bench.make_opt(cfg)
    if batch is not None:
        b = batch
        opt.batch_size = b
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(bench.make_clip(opt, cfg, b, frames, side, seed=1234))
    pg = dict(ts.model.named_parameters())
    name = next(k for k in pg if 'netG' in k and k.endswith('weight_orig'))
    p0 = pg[name].detach().clone()
    out, GL, DL = ts.gd_step(clip)
    assert set(GL) >= {'GAN_Img', 'GAN_Feat', 'loss_F_Warp', 'total_loss'} and set(DL) >= {'D_img_fake', 'D_img_real', 'total_img_loss'}
    assert all(bool(torch.isfinite(v).all()) for v in list(GL.values()) + list(DL.values())), (GL, DL)
    assert tuple(out[0].shape) == (b, frames, 3, side, side)
    assert bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[2]).all())
    assert (out[0][:, 0] == clip[0][:, 0]).all()                     # frame 0 is the ground-truth frame (generator.py:56)
    assert (pg[name].detach() - p0).abs().max().item() > 0
    gclip = ts.to_device(bench.make_clip(opt, cfg, b, frames, side, seed=4321))
    _, GG = ts.graph_step(gclip)
    assert bool(torch.isfinite(GG['total_loss']))
    return opt


def test_c2_step_runs_and_is_finite(device):
    """c2: CATER 128x128, 8 frames, batch 4, default widths."""
    _finite_step('c2', device)


def test_c5_shaped_step_runs_and_is_finite(device):
    """c5-shaped: Something-Else vocabulary (one attribute => semantic_nc 128, 2-3 objects incl. the hand), 224x224, 16 frames;
    per-GPU batch 2 keeps the test short (bench.py --config c5 runs batch 8)."""
    opt = _finite_step('c5', device, batch=2)
    assert opt.semantic_nc == 128 and tuple(opt.image_size) == (224, 224)
