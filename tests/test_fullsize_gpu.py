"""Parity at BASELINE.json's full (c3) layer sizes through size-independent properties: the oracle cannot run a 256x256x512
convolution in seconds, but (a) a convolution is local -- a window of the full-size GPU output must equal the oracle's result on
the matching crop; (b) forward / data-gradient / weight-gradient are adjoints of one bilinear form (dot-product test);
(c) normalisation outputs have known per-channel moments; (d) the pyramid adjoint conserves mass; (e) the warp is linear in the image
and maps constants to themselves;
(f) the two arithmetic modes of the convolution agree.  Tolerances are stated per check (fp32 with 1e-5-level split products)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, device='cpu', scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


# the three GEMM shapes of the SPADE stack at 256x256 (c3): mlp_shared 512 -> 128, gamma|beta 128 -> 2*64, conv_0 of up_3 128 -> 64
FULL = [(8, 512, 256, 256, 128), (8, 128, 256, 256, 128), (8, 128, 256, 256, 64)]


@pytest.mark.parametrize("shape", FULL, ids=str)
@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_conv_window_matches_oracle_crop(shape, prec, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Ko = shape
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        with ops.bf16x3_min_rows(48):
            x = _rand(N, C, H, W, seed=1, device=device)
            w = _rand(Ko, C, 3, 3, seed=2, device=device, scale=(C * 9) ** -0.5)
            b = _rand(Ko, seed=3, device=device)
            y = ops.conv2d(x, w, b, 1, 1)
        for (n, y0, x0) in [(0, 0, 0), (N - 1, H - 24, W - 24), (3, 101, 57)]:     # corner windows exercise the padding taps
            ys, xs = max(0, y0 - 1), max(0, x0 - 1)
            ye, xe = min(H, y0 + 25), min(W, x0 + 25)
            crop = x[n:n + 1, :, ys:ye, xs:xe].cpu()
            # pad the crop where the window touches the image border so that the oracle sees the same zero padding
            pt, pl = 1 - (y0 - ys), 1 - (x0 - xs)
            pb, pr = 1 - (ye - (y0 + 24)), 1 - (xe - (x0 + 24))
            crop = torch.nn.functional.pad(crop, (pl, pr, pt, pb))
            ref = K.conv2d(crop, w.cpu(), b.cpu(), 1, 0)                      # [1, Ko, 24, 24]
            got = y[n:n + 1, :, y0:y0 + 24, x0:x0 + 24].cpu()
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-4, (shape, prec, (n, y0, x0), err)
    finally:
        ops.set_conv_precision(old)


@pytest.mark.parametrize("shape", FULL[:2], ids=str)
def test_conv_adjoint_identities(shape, device):
    """<conv(x, w), dy> == <x, dgrad(dy)> == <w, wgrad(dy)> at full size (fp64 dot products of the fp32 kernel outputs)."""
    from ag2video_amd import ops
    N, C, H, W, Ko = shape
    x = _rand(N, C, H, W, seed=1, device=device).requires_grad_(True)
    w = _rand(Ko, C, 3, 3, seed=2, device=device, scale=(C * 9) ** -0.5).requires_grad_(True)
    dy = _rand(N, Ko, H, W, seed=3, device=device)
    y = ops.conv2d(x, w, None, 1, 1)
    y.backward(dy)
    a = (y.detach().double() * dy.double()).sum().item()
    bx = (x.detach().double() * x.grad.double()).sum().item()
    bw = (w.detach().double() * w.grad.double()).sum().item()
    scale = (y.detach().double().pow(2).sum().sqrt() * dy.double().pow(2).sum().sqrt()).item()
    assert abs(a - bx) <= 2e-5 * scale, (a, bx, scale)
    assert abs(a - bw) <= 2e-5 * scale, (a, bw, scale)


def test_conv_modes_agree_full_size(device):
    from ag2video_amd import ops
    N, C, H, W, Ko = FULL[0]
    x = _rand(N, C, H, W, seed=1, device=device)
    w = _rand(Ko, C, 3, 3, seed=2, device=device, scale=(C * 9) ** -0.5)
    old = ops.get_conv_precision()
    try:
        ops.set_conv_precision('fp32')
        y0 = ops.conv2d(x, w, None, 1, 1)
        ops.set_conv_precision('bf16x3')
        y1 = ops.conv2d(x, w, None, 1, 1)
    finally:
        ops.set_conv_precision(old)
    err = ((y0 - y1).abs().max() / y0.abs().max()).item()
    assert err <= 5e-5, err


def test_spade_norm_moments_full_size(device):
    """gamma = beta = 0, slope 1: the output is the batch-normalised input -> per-channel mean 0, variance 1."""
    from ag2video_amd import ops
    N, C, H, W = 8, 64, 256, 256
    x = _rand(N, C, H, W, seed=4, device=device) * 3.0 + 1.5
    gb = torch.zeros(N, 2 * C, H, W, device=device)
    rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
    y = ops.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, 1.0)
    m = y.double().mean(dim=(0, 2, 3))
    v = y.double().var(dim=(0, 2, 3), unbiased=False)
    assert m.abs().max().item() <= 1e-4 and (v - 1).abs().max().item() <= 1e-4, (m.abs().max().item(), (v - 1).abs().max().item())
    # running statistics: momentum 0.1 of the batch mean / unbiased variance
    xm = x.double().mean(dim=(0, 2, 3))
    assert ((rm.double() - 0.1 * xm).abs().max() <= 1e-5 * xm.abs().max()).item()


def test_pyramid_adjoint_conserves_mass_full_size(device):
    from ag2video_amd import ops
    N, C, H = 8, 512, 256
    x = _rand(N, C, H, H, seed=5, device=device).requires_grad_(True)
    sizes = [(8 << i, 8 << i) for i in range(6)]
    ys = ops.nearest_pyramid(x, sizes)
    gs = [_rand(*y.shape, seed=10 + i, device=device) for i, y in enumerate(ys)]
    torch.autograd.backward(ys, gs)
    total = sum(g.double().sum().item() for g in gs)
    got = x.grad.double().sum().item()
    norm = sum(g.double().abs().sum().item() for g in gs)
    assert abs(got - total) <= 1e-6 * norm, (got, total)
    # and level values land on the grid points: dx at (0, 0) carries one term from every level
    want00 = sum(g[0, 0, 0, 0].item() for g in gs)
    assert abs(x.grad[0, 0, 0, 0].item() - want00) <= 1e-5 * max(1.0, abs(want00))


def test_flow_warp_is_linear_and_preserves_constants_full_size(device):
    """Bilinear border-clamped sampling is linear in the image and maps a constant image to itself for ANY flow.  (A zero flow
    is NOT the identity here: the reference builds its grid with linspace(-1, 1) but samples with align_corners=False,
    models/utils.py:113-140 -- the product reproduces that.)"""
    from ag2video_amd import ops
    a, b = _rand(8, 3, 256, 256, seed=6, device=device), _rand(8, 3, 256, 256, seed=7, device=device)
    flow = _rand(8, 2, 256, 256, seed=8, device=device) * 6.0
    lhs = ops.flow_warp(0.7 * a - 1.3 * b, flow)
    rhs = 0.7 * ops.flow_warp(a, flow) - 1.3 * ops.flow_warp(b, flow)
    assert (lhs - rhs).abs().max().item() <= 1e-5 * max(1.0, rhs.abs().max().item())
    const = torch.full((8, 3, 256, 256), 0.375, device=device)
    assert (ops.flow_warp(const, flow) - const).abs().max().item() <= 1e-6


def test_full_size_step_runs_and_is_finite(device):
    """One c3 training step (256x256, 16 frames, batch 8): finite losses of the documented keys, parameters move."""
    import bench
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = bench.make_opt('c3')
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
    pg = dict(ts.model.named_parameters())
    name = next(k for k in pg if 'netG' in k and k.endswith('weight_orig'))     # a generator weight (the graph nets have their own step)
    p0 = pg[name].detach().clone()
    out, GL, DL = ts.gd_step(clip)
    assert set(GL) >= {'GAN_Img', 'GAN_Feat', 'loss_F_Warp', 'total_loss'} and set(DL) >= {'D_img_fake', 'D_img_real', 'total_img_loss'}
    assert all(bool(torch.isfinite(v).all()) for v in list(GL.values()) + list(DL.values()))
    assert tuple(out[0].shape) == (batch, frames, 3, side, side)
    assert (pg[name].detach() - p0).abs().max().item() > 0


def test_full_size_frame_batched_equals_sequential(device):
    """c3 (256x256, 16 frames, batch 8, default widths): the frame-batched generator pass (one [120, ...] batch, per-group statistics
    and sigma_t, 16 GB tensors convolved in sample chunks) against the sequential 15-pass loop the golden vectors pin -- same weights,
    same clip.  This test found a real bug (the few-row kernel's padding taps read live data beyond 256 MB of input: O(1) errors from
    frame 3 on).  Tolerances: the flows (exact-fp32 flow network) 1e-4; the images 1e-2 -- at these widths and 256x256 the randomly
    initialised decoder amplifies last-bit differences (the order of split-K atomics) to 1.5-2e-3 of max between two runs of the SAME
    path (tools/debug_batched_vs_seq.py PAIR=ss / bb), so 1e-2 separates "same arithmetic" from a plumbing error (O(0.1 - 1));
    the running statistics and spectral-norm vectors after the 15 per-pass updates 1e-4."""
    import os
    import bench
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = bench.make_opt('c3')
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
    sd0 = {k: v.clone() for k, v in ts.model.state_dict().items()}
    outs, states = [], []
    for seq in (False, True):
        ts.model.load_state_dict(sd0)
        opt.batch_frames = not seq
        with torch.no_grad():
            o = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
        outs.append([t.clone() for t in o[:4]])
        states.append({k: v.clone() for k, v in ts.model.state_dict().items() if 'running_' in k or k.endswith('weight_u') or k.endswith('weight_v')})
        del o
    opt.batch_frames = True
    for name, tol, a, b in zip(('imgs_pred', 'boxes_pred', 'flows_pred'), (1e-2, 1e-6, 1e-4), outs[0], outs[1]):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err <= tol, (name, err)
        for f in range(a.shape[1]):                     # every frame group on its own (a chunk-boundary bug hits late groups only)
            ef = (a[:, f] - b[:, f]).abs().max().item() / b.abs().max().item()
            assert ef <= tol, (name, f, ef)
    assert (outs[0][3] != outs[1][3]).float().mean().item() < 1e-3          # confidence masks
    for k in states[1]:
        a, b = states[0][k], states[1][k]
        assert (a - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1e-6), k


def test_fewrow_conv_beyond_256mb(device):
    """<= 4 output rows (conv_img, conv_flow, the discriminators' last layer) on an input larger than the 256 MB one launch of the
    few-row kernel can address: the full batch must equal the same convolution run on sample slices, and a corner window the oracle."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H = 20, 64, 256                                    # 335 MB
    x = _rand(N, C, H, H, seed=1, device=device)
    w = _rand(3, C, 3, 3, seed=2, device=device, scale=(C * 9) ** -0.5)
    b = _rand(3, seed=3, device=device)
    y = ops.conv2d(x, w, b, 1, 1)
    parts = torch.cat([ops.conv2d(x[i:i + 4], w, b, 1, 1) for i in range(0, N, 4)], dim=0)
    assert torch.equal(y, parts)
    for n in (0, N - 1):
        crop = torch.nn.functional.pad(x[n:n + 1, :, :25, :25].cpu(), (1, 0, 1, 0))
        ref = K.conv2d(crop, w.cpu(), b.cpu(), 1, 0)
        got = y[n:n + 1, :, :24, :24].cpu()
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), n


def test_full_size_step_gradients_batched_equal_sequential(device):
    """The BACKWARD of the frame-batched pass at c3 (grouped weight gradients + sn_group_combine, chunked data / weight gradients,
    per-group batch-norm backward) against the sequential loop's, same state and clip: every generator parameter's gradient (abs-sum
    and L2) within 5e-2 (the forward already differs by ~2e-3 between two runs of one path at these sizes, see above; L1 / hinge /
    LeakyReLU / warp derivatives are sign functions), the loss values within 1e-2.  A chunk-boundary or group-indexing error shows up as
    an O(1) deviation of the affected layers."""
    import bench
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = bench.make_opt('c3')
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
    sdG = {k: v.clone() for k, v in ts.model.state_dict().items()}
    sdD = {k: v.clone() for k, v in ts.discriminator.img_discriminator.state_dict().items()}
    res = []
    for seq in (False, True):
        ts.model.load_state_dict(sdG)
        ts.discriminator.img_discriminator.load_state_dict(sdD)
        opt.batch_frames = not seq
        _, GL, DL = ts.gd_step(clip, apply_updates=False)
        grads = {n: (p.grad.double().abs().sum().item(), p.grad.double().pow(2).sum().sqrt().item(), p.numel())
                 for n, p in ts.model.named_parameters() if p.grad is not None and not n.startswith('acts_to_boxes')}
        res.append(({k: v.item() for k, v in GL.items()}, grads))
    opt.batch_frames = True
    (lb, gb), (ls, gs) = res
    for k in ls:
        assert abs(lb[k] - ls[k]) <= 1e-2 * max(1.0, abs(ls[k])), (k, lb[k], ls[k])
    assert set(gb) == set(gs) and len(gs) > 250
    bad = []
    for n in gs:
        for i in (0, 1):
            floor = 2e-6 * gs[n][2] if i == 0 else 2e-6 * gs[n][2] ** 0.5
            if abs(gb[n][i] - gs[n][i]) > 5e-2 * gs[n][i] + floor:
                bad.append((n, i, gb[n][i], gs[n][i]))
    assert not bad, (len(bad), bad[:8])


def test_full_size_deterministic_mode(device):
    """c3 (256x256, 16 frames, batch 8) with ops.deterministic() (without it two runs of the same binary differ by
    1.5-2e-3 of max|image| through the order of split-K atomics, which capped every full-size check at 1e-2):
    (a) two runs of the frame-batched forward and of the whole G + D step from the same state are BIT-identical -- outputs, losses,
        every gradient, the spectral-norm vectors and running statistics;
    (b) the frame-batched pass against the sequential 15-pass loop: both deterministic, so what is left is the rounding difference of
        their two formulations (conv(x, W) * (1/sigma_t) in the epilogue vs conv(x, W / sigma_t); statistics of 8 samples inside a
        120-sample launch) amplified by the randomly initialised decoder -- measured values are printed, the bounds are the
        tightened ones the verdict asks for where the formulations allow it."""
    import bench
    from ag2video_amd import ops
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = bench.make_opt('c3')
    opt.gpu_ids = [0]
    with ops.deterministic():
        ts = TrainStep(opt, device, weights_seed=0)
        assert ts.side_stream is None
        clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
        sdG = {k: v.clone() for k, v in ts.model.state_dict().items()}
        sdD = {k: v.clone() for k, v in ts.discriminator.img_discriminator.state_dict().items()}

        def fwd(seq):
            ts.model.load_state_dict(sdG)
            opt.batch_frames = not seq
            with torch.no_grad():
                o = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
            st = {k: v.clone() for k, v in ts.model.state_dict().items() if 'running_' in k or k.endswith('weight_u') or k.endswith('weight_v')}
            return [t.clone() for t in o[:3]], st

        def step(seq):
            ts.model.load_state_dict(sdG)
            ts.discriminator.img_discriminator.load_state_dict(sdD)
            opt.batch_frames = not seq
            _, GL, DL = ts.gd_step(clip, apply_updates=False)
            torch.cuda.synchronize()
            return ({k: v.item() for k, v in list(GL.items()) + list(DL.items())}, ts.optimizer_generator.flat_grad.clone(),
                    ts.discriminator.optimizer_d_img.flat_grad.clone())
        try:
            (o0, s0), (o1, s1) = fwd(False), fwd(False)
            for a, b in zip(o0, o1):
                assert torch.equal(a, b), "frame-batched forward: two deterministic runs differ"
            assert all(torch.equal(s0[k], s1[k]) for k in s0)
            o2, s2 = fwd(True)
            errs = {n: ((a - b).abs().max() / b.abs().max()).item() for n, a, b in zip(('imgs_pred', 'boxes_pred', 'flows_pred'), o0, o2)}
            serr = max(((s0[k] - s2[k]).abs().max() / s2[k].abs().max().clamp(min=1e-6)).item() for k in s2)
            print('deterministic c3, batched vs sequential forward (error / max):', errs, 'state', serr)
            del o0, o1, o2
            l0, gG0, gD0 = step(False)
            l1, gG1, gD1 = step(False)
            assert l0 == l1, (l0, l1)
            assert torch.equal(gG0, gG1) and torch.equal(gD0, gD1), "G + D step: two deterministic runs differ"
            del gG1, gD1
            l2, gG2, gD2 = step(True)
            lerr = max(abs(l0[k] - l2[k]) / max(1.0, abs(l2[k])) for k in l2)
            gerr = (((gG0 - gG2).norm() / gG2.norm()).item(), ((gD0 - gD2).norm() / gD2.norm()).item())
            print('deterministic c3, batched vs sequential step: losses', lerr, 'gradients (relative L2: G, D)', gerr)
        finally:
            opt.batch_frames = True
    # Measured (r03): flows and boxes bit-identical, state 1e-6, images 6.2e-4 of max, losses identical, gradients 1.1e-3 (G) / 2e-4 (D)
    # in relative L2.  At batch 2 the two passes agree BIT FOR BIT in every decoder block (tools/debug_det_paths.py); at batch 8 they
    # part ways at the 1e-6 level in front of netG.fc: the low-rank layout taps come from a small fp32 GEMM whose kernel (K order)
    # is chosen by its row count (120 x 22 object slots batched, 8 x 22 per pass), and the decoder amplifies that ~300x.  The bounds
    # are 5-10x tighter than the non-deterministic tests above (1e-2 / 5e-2).
    # (Round 4: the halo-tile fp16-split kernel, which only launches of >= 384 tiles take, stays off in deterministic mode -- with it the
    # batched pass's flows differ from the sequential passes' by 2.1e-6 of max and the random-init decoder turns that into 1.4e-2 of the
    # generator gradient: what this test is there to catch.)
    assert errs['flows_pred'] <= 1e-6 and errs['boxes_pred'] <= 1e-6 and serr <= 1e-5, (errs, serr)
    assert errs['imgs_pred'] <= 2e-3, errs
    assert lerr <= 1e-5 and gerr[0] <= 5e-3 and gerr[1] <= 1e-3, (lerr, gerr)
