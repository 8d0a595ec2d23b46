"""Parity of every HIP operator (through the C ABI, via ag2video_amd.ops) against the CPU oracle (oracle/kernels.py),
forward and backward, on seeded inputs.

Tolerance (stated once): max|got - ref| <= TOL * max|ref| with TOL = 2e-4 for every tensor compared -- an order tighter
than BASELINE.json's 1e-3 relative fp32 bound, so that errors cannot pile up across the ~100 layers of the step.
"""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _cmp(name, got, ref, tol=TOL):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "%s: non-finite output" % name
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-20
    assert err <= tol * scale, "%s: max err %.3e, ref scale %.3e (rel %.2e)" % (name, err, scale, err / scale)


def _run_pair(fn_gpu, fn_cpu, tensors, device, nondiff=(), name='op', tol=TOL, out_weights_seed=3):
    """tensors: dict name -> cpu tensor (float tensors get grads unless listed in nondiff)."""
    cpu_in, gpu_in = {}, {}
    for k, t in tensors.items():
        if torch.is_tensor(t):
            diff = t.is_floating_point() and k not in nondiff
            cpu_in[k] = t.clone().requires_grad_(diff)
            gpu_in[k] = t.clone().to(device).requires_grad_(diff)
        else:
            cpu_in[k] = gpu_in[k] = t
    ref = fn_cpu(**cpu_in)
    got = fn_gpu(**gpu_in)
    _cmp(name + '.fwd', got, ref, tol)
    g = torch.Generator().manual_seed(out_weights_seed)
    wgt = torch.randn(ref.shape, generator=g)
    (ref * wgt).sum().backward()
    (got * wgt.to(device)).sum().backward()
    for k in tensors:
        if torch.is_tensor(cpu_in[k]) and cpu_in[k].requires_grad:
            assert gpu_in[k].grad is not None, "%s: no grad for %s" % (name, k)
            _cmp(name + '.grad_' + k, gpu_in[k].grad, cpu_in[k].grad, tol)
    return got, ref


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # N, C, H, W, K, R, S, stride, pad, bias, residual
    (2, 5, 9, 11, 7, 3, 3, 1, 1, True, False),
    (2, 16, 16, 16, 40, 3, 3, 1, 1, True, True),
    (1, 24, 20, 20, 130, 3, 3, 1, 1, False, False),
    (2, 19, 17, 17, 8, 4, 4, 2, 2, True, False),
    (2, 8, 9, 9, 16, 4, 4, 1, 2, True, False),
    (2, 32, 8, 8, 16, 1, 1, 1, 0, False, False),
    (1, 1027, 16, 16, 32, 3, 3, 1, 1, False, False),
    (2, 8, 16, 16, 16, 3, 3, 2, 1, False, False),
    (4, 64, 64, 64, 64, 3, 3, 1, 1, True, False),
    (3, 259, 33, 33, 64, 4, 4, 2, 2, True, False),
    (2, 130, 12, 12, 3, 3, 3, 1, 1, True, False),     # <= 4 output channels: few-row forward and weight-gradient kernels
    (2, 20, 11, 13, 1, 4, 4, 1, 2, True, False),      # one output channel, 4x4 pad 2 (the discriminator's last layer)
    (2, 3, 16, 16, 200, 3, 3, 1, 1, True, False),     # <= 4 input channels: data gradient as 1x1 GEMM + shifted-plane sum
    (2, 2, 10, 12, 72, 4, 4, 1, 2, False, False),     # same path, 4x4 filter with pad 2 (output one larger than the input)
    (2, 64, 8, 8, 48, 3, 3, 1, 1, True, True),        # few pixels, long K loop: split-K (atomic epilogue) path
    (4, 128, 16, 16, 200, 3, 3, 1, 1, True, False),
    (2, 40, 9, 9, 150, 1, 1, 2, 0, False, False),     # 1x1 stride 2: three of the four parity classes are empty
    (2, 12, 20, 18, 24, 3, 3, 2, 1, True, False),
    # >= 128 output channels, stride 1, OW % 8 == 0: weight gradient through the 16-byte-load kernel (runs of 8 pixels)
    (2, 24, 8, 24, 136, 3, 3, 1, 1, True, False),     # H != W
    (2, 12, 7, 15, 130, 4, 4, 1, 2, False, False),    # pad 2: two leading / one trailing element of a border run outside
    (3, 16, 8, 8, 256, 1, 1, 1, 0, False, False),     # 1x1
    (2, 10, 12, 10, 128, 3, 3, 1, 0, True, False),    # pad 0, OW = 8 < W
    # gathered channels % 32 == 0, stride 1, OW % 4 == 0: tap-major producers with quad (16-byte) loads of the gathered operand
    (2, 64, 12, 16, 136, 3, 3, 1, 1, True, False),    # forward on the 128 x 256 tile
    (2, 128, 10, 12, 160, 3, 3, 1, 1, False, True),   # forward and data gradient (160 gathered channels)
    (2, 32, 9, 11, 128, 4, 4, 1, 2, True, False),     # 4x4, pad 2: two border elements per edge quad
    (1, 96, 8, 8, 256, 3, 3, 1, 1, False, False),     # 256-row tile, the quad before the tensor's first element (x = -1 at row 0)
    (1, 5, 40, 40, 192, 3, 3, 1, 1, False, True),     # pixel count not a multiple of the K-tile of a split
    # <= 4 output rows, stride 1: the few-row FMA kernel (forward here; data gradient for <= 4 input channels)
    (3, 20, 9, 9, 1, 4, 4, 1, 2, True, False),        # the discriminators' last layer: OW = 10, not a multiple of 4
    (2, 16, 8, 12, 2, 1, 1, 1, 0, True, True),
    (2, 4, 11, 13, 4, 3, 3, 1, 1, False, False),
]


@pytest.fixture(params=['fp32', 'bf16x3'])
def precision(request):
    """Both arithmetic modes of the 128-row conv tiles are held to the same bound (2e-4 of max|ref|)."""
    from ag2video_amd import ops
    old = ops.get_conv_precision()
    ops.set_conv_precision(request.param)
    yield request.param
    ops.set_conv_precision(old)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d(case, precision, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Ko, R, S, st, pad, bias, res = case
    x = _rand(N, C, H, W, seed=1)
    w = _rand(Ko, C, R, S, seed=2, scale=(C * R * S) ** -0.5)
    t = {'x': x, 'w': w, 'bias': _rand(Ko, seed=3) if bias else None, 'stride': st, 'pad': pad}
    if res:
        OH, OW = (H + 2 * pad - R) // st + 1, (W + 2 * pad - S) // st + 1
        t['residual'] = _rand(N, Ko, OH, OW, seed=4)
    else:
        t['residual'] = None
    _run_pair(ops.conv2d, K.conv2d, t, device, name='conv2d')


@pytest.mark.parametrize("case", [(2, 24, 16, 16, 136, 3, 1, 1), (2, 130, 8, 8, 160, 3, 1, 1), (1, 20, 12, 12, 40, 3, 1, 1),
                                  (2, 6, 9, 9, 3, 3, 1, 1), (2, 16, 10, 10, 24, 4, 2, 1)], ids=str)
def test_conv2d_relu_pair(case, precision, device):
    """SPADE's pattern: a = relu(conv(x, w1)) with the ReLU in the epilogue, y = conv(a, w2) whose data gradient applies the
    ReLU's backward (relu=True / grad_premasked=True on the producer, relu_input=True on the consumer) -- against the oracle."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Km, R, st, pad = case
    t = {'x': _rand(N, C, H, W, seed=1), 'w1': _rand(Km, C, R, R, seed=2, scale=(C * R * R) ** -0.5), 'b1': _rand(Km, seed=3),
         'w2': _rand(2 * C, Km, 3, 3, seed=4, scale=(Km * 9) ** -0.5)}

    def run(mod):
        def f(x, w1, b1, w2):
            a = mod.conv2d(x, w1, b1, st, pad, relu=True, grad_premasked=True)
            return mod.conv2d(a, w2, None, 1, 1, relu_input=True)
        return f
    _run_pair(run(ops), run(K), t, device, name='conv_relu_pair')
    # and the stand-alone epilogue ReLU (its own backward masks dy)
    t2 = {'x': t['x'], 'w': t['w1'], 'bias': t['b1']}
    _run_pair(lambda x, w, bias: ops.conv2d(x, w, bias, st, pad, relu=True), lambda x, w, bias: K.conv2d(x, w, bias, st, pad, relu=True),
              t2, device, name='conv_relu')


@pytest.mark.parametrize("case", [(2, 24, 16, 16, (136, 136, 130), 3, 1, 1), (1, 130, 8, 8, (160, 40), 3, 1, 1),
                                  (2, 12, 9, 11, (24, 3, 200), 3, 1, 1), (2, 16, 10, 10, (136, 24), 4, 2, 1)], ids=str)
def test_conv2d_shared_input(case, precision, device):
    """Several convolutions of one input with the in-place accumulated input gradient (ops.conv2d_shared_input)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Ks, R, st, pad = case
    x = _rand(N, C, H, W, seed=1)
    ws = [_rand(k, C, R, R, seed=10 + i, scale=(C * R * R) ** -0.5) for i, k in enumerate(Ks)]
    bs = [_rand(k, seed=20 + i) for i, k in enumerate(Ks)]

    def run(mod, dev):
        xx = x.clone().to(dev).requires_grad_(True)
        wl = [w.clone().to(dev).requires_grad_(True) for w in ws]
        bl = [b.clone().to(dev).requires_grad_(True) for b in bs]
        ys = mod.conv2d_shared_input(xx, wl, bl, st, pad, relu=True)
        loss = 0.0
        for i, y in enumerate(ys):
            loss = loss + (y * _rand(*y.shape, seed=30 + i).to(dev)).sum() * (i + 1)
        loss.backward()
        return ys, xx.grad, [w.grad for w in wl], [b.grad for b in bl]

    yg, dxg, dwg, dbg = run(ops, device)
    yr, dxr, dwr, dbr = run(K, 'cpu')
    for i in range(len(Ks)):
        _cmp('shared.fwd%d' % i, yg[i], yr[i], TOL)
        _cmp('shared.dw%d' % i, dwg[i], dwr[i], TOL)
        _cmp('shared.db%d' % i, dbg[i], dbr[i], TOL)
    _cmp('shared.dx', dxg, dxr, TOL)


@pytest.mark.parametrize("shape", [(40, 16, 3, 3), (130, 1027, 3, 3), (16, 8, 4, 4), (8, 16, 1, 1)])
@pytest.mark.parametrize("training", [True, False])
def test_spectral_weight(shape, training, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    w = _rand(*shape, seed=5, scale=0.1)
    M, R = shape[0], shape[1] * shape[2] * shape[3]
    u = torch.nn.functional.normalize(_rand(M, seed=6), dim=0)
    v = torch.nn.functional.normalize(_rand(R, seed=7), dim=0)
    uc, vc = u.clone(), v.clone()
    ug, vg = u.clone().to(device), v.clone().to(device)
    wc = w.clone().requires_grad_(True)
    wg = w.clone().to(device).requires_grad_(True)
    ref = K.spectral_weight(wc, uc, vc, training)
    got = ops.spectral_weight(wg, ug, vg, training)
    _cmp('sn.fwd', got, ref)
    _cmp('sn.u', ug, uc)
    _cmp('sn.v', vg, vc)
    wgt = _rand(*shape, seed=8)
    (ref * wgt).sum().backward()
    (got * wgt.to(device)).sum().backward()
    _cmp('sn.grad_w', wg.grad, wc.grad)
    # a second call keeps iterating on the updated buffers
    ref2 = K.spectral_weight(wc, uc, vc, training)
    got2 = ops.spectral_weight(wg, ug, vg, training)
    _cmp('sn.fwd2', got2, ref2)


NORM_SHAPES = [(2, 8, 2, 2), (2, 16, 8, 8), (3, 5, 7, 9), (2, 4, 96, 96), (4, 33, 64, 64)]


@pytest.mark.parametrize("shape", NORM_SHAPES, ids=str)
@pytest.mark.parametrize("slope", [0.2, 1.0])
def test_batchnorm_act(shape, slope, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    C = shape[1]
    x = _rand(*shape, seed=1) * 1.7 + 0.6
    t = {'x': x, 'weight': 1 + 0.1 * _rand(C, seed=2), 'bias': 0.1 * _rand(C, seed=3)}
    rm_c, rv_c = torch.zeros(C), torch.ones(C)
    rm_g, rv_g = torch.zeros(C, device=device), torch.ones(C, device=device)
    _run_pair(lambda **k: ops.batchnorm_act(k['x'], k['weight'], k['bias'], rm_g, rv_g, True, 0.1, 1e-5, slope),
              lambda **k: K.batchnorm_act(k['x'], k['weight'], k['bias'], rm_c, rv_c, True, 0.1, 1e-5, slope),
              t, device, name='bn_act')
    _cmp('bn.running_mean', rm_g, rm_c)
    _cmp('bn.running_var', rv_g, rv_c)


@pytest.mark.parametrize("shape", NORM_SHAPES, ids=str)
@pytest.mark.parametrize("slope", [0.2, 1.0])
def test_spade_norm(shape, slope, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    C = shape[1]
    t = {'x': _rand(*shape, seed=1) * 1.3 - 0.4, 'gamma': 0.5 * _rand(*shape, seed=2), 'beta': 0.5 * _rand(*shape, seed=3)}
    rm_c, rv_c = torch.zeros(C), torch.ones(C)
    rm_g, rv_g = torch.zeros(C, device=device), torch.ones(C, device=device)
    _run_pair(lambda **k: ops.spade_norm(k['x'], k['gamma'], k['beta'], rm_g, rv_g, True, 0.1, 1e-5, slope),
              lambda **k: K.spade_norm(k['x'], k['gamma'], k['beta'], rm_c, rv_c, True, 0.1, 1e-5, slope),
              t, device, name='spade')
    _cmp('spade.running_mean', rm_g, rm_c)
    _cmp('spade.running_var', rv_g, rv_c)


@pytest.mark.parametrize("shape", NORM_SHAPES, ids=str)
@pytest.mark.parametrize("slope", [0.2, 1.0])
def test_spade_norm_fused(shape, slope, device):
    """gamma | beta read in place from one [N, 2C, H, W] tensor (the fused mlp_gamma / mlp_beta convolution's output)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C = shape[0], shape[1]
    t = {'x': _rand(*shape, seed=1) * 1.3 - 0.4, 'gb': 0.5 * _rand(N, 2 * C, *shape[2:], seed=2)}
    rm_c, rv_c = torch.zeros(C), torch.ones(C)
    rm_g, rv_g = torch.zeros(C, device=device), torch.ones(C, device=device)
    _run_pair(lambda **k: ops.spade_norm_fused(k['x'], k['gb'], rm_g, rv_g, True, 0.1, 1e-5, slope),
              lambda **k: K.spade_norm_fused(k['x'], k['gb'], rm_c, rv_c, True, 0.1, 1e-5, slope),
              t, device, name='spade_fused')
    _cmp('spade_fused.running_mean', rm_g, rm_c)
    _cmp('spade_fused.running_var', rv_g, rv_c)


def test_norm_eval_mode(device):
    from ag2video_amd import ops
    from oracle import kernels as K
    x = _rand(2, 6, 8, 8, seed=1)
    rm, rv = 0.2 * _rand(6, seed=2), 1 + 0.3 * torch.rand(6, generator=torch.Generator().manual_seed(3))
    w, b = 1 + 0.1 * _rand(6, seed=4), 0.1 * _rand(6, seed=5)
    ref = K.batchnorm_act(x, w, b, rm.clone(), rv.clone(), False, 0.1, 1e-5, 0.2)
    got = ops.batchnorm_act(x.to(device), w.to(device), b.to(device), rm.to(device), rv.to(device), False, 0.1, 1e-5, 0.2)
    _cmp('bn.eval', got, ref)


@pytest.mark.parametrize("shape", [(2, 8, 9, 9), (3, 16, 33, 33), (2, 4, 129, 129), (5, 7, 6, 5)], ids=str)
def test_instnorm_lrelu(shape, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    t = {'x': _rand(*shape, seed=1) * 2 + 0.3}
    _run_pair(ops.instnorm_lrelu, K.instnorm_lrelu, t, device, name='instnorm')


def _boxes(BT, O, seed):
    g = torch.Generator().manual_seed(seed)
    w = 0.094 + (0.22 - 0.094) * torch.rand(BT, O, generator=g)
    h = 0.125 + (0.29 - 0.125) * torch.rand(BT, O, generator=g)
    x0 = torch.rand(BT, O, generator=g) * (1 - w)
    y0 = torch.rand(BT, O, generator=g) * (1 - h)
    return torch.stack([x0, y0, w, h], dim=-1)


@pytest.mark.parametrize("cfg", [(3, 5, 16, 32, 32), (4, 11, 40, 64, 64), (2, 7, 520, 24, 40)], ids=str)
def test_layout_render(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    BT, O, D, H, W = cfg
    boxes = _boxes(BT, O, 11)
    boxes[0, 1] = 0.0                       # an all-zero ("illegal") box is dropped (layout.py:40-42)
    boxes[1, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0])   # full-image box
    boxes[BT - 1, O - 1] = -1.0             # collate padding
    keep = torch.ones(BT, O, dtype=torch.int32)
    keep[:, O - 1] = 0
    keep[0, 2] = 0
    t = {'vecs': _rand(BT, O, D, seed=12), 'boxes': boxes, 'keep': keep, 'H': H, 'W': W}
    _run_pair(ops.layout_render, K.layout_render, t, device, nondiff=('boxes',), name='layout')


@pytest.mark.parametrize("cfg", [(2, 6, 20, 32, 32, 3, 1, 1, 0), (3, 22, 40, 64, 64, 3, 1, 1, 0), (2, 5, 9, 32, 32, 4, 2, 2, 0),
                                 (1, 9, 33, 24, 40, 3, 1, 1, 0), (3, 7, 12, 64, 64, 4, 2, 2, 1), (2, 4, 8, 33, 29, 4, 2, 2, 1),
                                 (2, 26, 20, 32, 32, 3, 1, 1, 0), (1, 24, 70, 16, 272, 3, 1, 1, 0), (2, 3, 200, 16, 16, 3, 1, 1, 0),
                                 # layout_conv_fwd3_kernel (3 x 3, K % 16 == 0, rows of whole quads): rows touching more objects than its tap
                                 # table holds, map heights that are not multiples of 4, the widest rows it takes, one channel block
                                 (3, 22, 48, 36, 64, 3, 1, 1, 0), (2, 24, 32, 18, 256, 3, 1, 1, 0), (2, 5, 16, 7, 12, 3, 1, 1, 0)],
                         ids=str)
def test_layout_conv(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    NB, J, Kc, H, W, R, st, pad, pool = cfg
    boxes = _boxes(NB, J, 21)
    boxes[0, 1] = 0.0
    boxes[NB - 1, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    boxes[0, J - 1] = -1.0
    if J >= 22 and Kc % 16 == 0:            # 14 tall objects in the last image: its rows touch more objects than fwd3's tap table holds
        boxes[NB - 1, 1:15, 1] = 0.02
        boxes[NB - 1, 1:15, 3] = 0.9
    keep = torch.ones(NB, J, dtype=torch.int32)
    keep[:, J - 1] = 0
    Hm, Wm = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool else (H, W)
    OH, OW = (Hm + 2 * pad - R) // st + 1, (Wm + 2 * pad - R) // st + 1
    t = {'U': _rand(NB, J, Kc, R * R, seed=22), 'boxes': boxes, 'keep': keep, 'H': H, 'W': W, 'R': R, 'stride': st, 'pad': pad,
         'residual': _rand(NB, Kc, OH, OW, seed=23), 'pool': pool}
    _run_pair(ops.layout_conv, K.layout_conv, t, device, nondiff=('boxes',), name='layout_conv')


def _sample_coords(flow, H, W):
    """Sampling coordinates of resample()/get_grid (models/utils.py:113-140) in fp64: grid = linspace(-1, 1) + flow / ((W-1)/2),
    unnormalised with align_corners=False: ix = ((g + 1) * W - 1) / 2 = (x + flow_x) * W / (W - 1) - 1/2."""
    f = flow.double()
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    return (xs + f[:, 0]) * W / (W - 1) - 0.5, (ys + f[:, 1]) * H / (H - 1) - 0.5


def _flow_for_targets(tx, ty, H, W):
    """Inverse of _sample_coords: the flow that makes pixel (y, x) sample the image at (ty, tx)."""
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    return torch.stack([(tx + 0.5) * (W - 1) / W - xs, (ty + 0.5) * (H - 1) / H - ys], dim=1).float()


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (3, 3, 33, 20), (2, 3, 64, 64)], ids=str)
@pytest.mark.parametrize("reach", [1.5, 9.0])
def test_flow_warp_off_cell_borders(shape, reach, device):
    """Sampling points constructed to stay >= 0.15 of a cell away from every cell border and from the clamp at the image border
    (inside: fraction in [0.15, 0.85]; outside: >= 0.65 beyond the edge, where border padding makes the derivative exactly zero).
    There d(warp)/d(flow) is continuous, so forward, image gradient AND flow gradient are held to the common 2e-4."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W = shape
    g = torch.Generator().manual_seed(7)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W).expand(N, H, W)
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1).expand(N, H, W)
    def targets(base, size):
        cell = torch.floor(base + (torch.rand(base.shape, generator=g, dtype=torch.float64) * 2 - 1) * reach)
        frac = 0.15 + 0.7 * torch.rand(base.shape, generator=g, dtype=torch.float64)
        t = cell + frac
        # cells straddling the clamp (ix in (-1, 0) or (size-1, size-0.5...)) are moved fully outside
        t = torch.where((t < 0), torch.minimum(t, torch.full_like(t, -0.65)), t)
        t = torch.where((t > size - 1), torch.maximum(t, torch.full_like(t, size - 1 + 0.65)), t)
        return t
    flow = _flow_for_targets(targets(xs, W), targets(ys, H), H, W)
    ix, iy = _sample_coords(flow, H, W)          # what the fp32 flow actually encodes: check the construction survived rounding
    for c, size in ((ix, W), (iy, H)):
        inside = (c > 0) & (c < size - 1)
        fr = c - torch.floor(c)
        assert bool(((fr[inside] > 0.1) & (fr[inside] < 0.9)).all()) and bool(((c[~inside] < -0.5) | (c[~inside] > size - 0.5)).all())
    t = {'img': _rand(N, C, H, W, seed=1), 'flow': flow}
    _run_pair(ops.flow_warp, K.flow_warp, t, device, name='warp')


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (3, 3, 33, 20), (2, 3, 64, 64)], ids=str)
@pytest.mark.parametrize("mag", [0.5, 6.0, 40.0])
def test_flow_warp(shape, mag, device):
    """Random flows (cell borders are crossed freely).  Forward and the image gradient are continuous in the flow: 2e-4.  The flow
    gradient is piecewise constant per sampling cell: it is compared at 2e-4 on every pixel whose sampling point is farther than
    1e-3 of a cell from a border (> 99 % of them) -- on the others the oracle's fp32 coordinate and the kernel's may fall into
    different cells, an O(1) difference that says nothing about either."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W = shape
    img, flow = _rand(N, C, H, W, seed=1), _rand(N, 2, H, W, seed=2) * mag
    ic, fc = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    ig, fg = img.clone().to(device).requires_grad_(True), flow.clone().to(device).requires_grad_(True)
    ref, got = K.flow_warp(ic, fc), ops.flow_warp(ig, fg)
    _cmp('warp.fwd', got, ref)
    wgt = _rand(*ref.shape, seed=3)
    (ref * wgt).sum().backward()
    (got * wgt.to(device)).sum().backward()
    _cmp('warp.grad_img', ig.grad, ic.grad)
    ix, iy = _sample_coords(flow, H, W)
    safe = torch.ones(N, H, W, dtype=torch.bool)
    for c, size in ((ix, W), (iy, H)):
        fr = c - torch.floor(c)
        safe &= ((fr > 1e-3) & (fr < 1 - 1e-3)) | (c < -1e-3) | (c > size - 1 + 1e-3)
        safe &= (c.abs() > 1e-3) & ((c - (size - 1)).abs() > 1e-3)
    assert safe.double().mean().item() > 0.99
    m = safe.unsqueeze(1).expand(N, 2, H, W)
    _cmp('warp.grad_flow[safe]', fg.grad.cpu() * m, fc.grad * m)


def test_flow_warp_on_cell_borders(device):
    """Border-crossing case: every sampling point sits EXACTLY on a grid node (integer coordinates, some outside the image).
    The forward value is then the image value at the clamped node whichever cell an implementation assigns the point to, and the
    image gradient is a pure scatter: both exact to rounding.  (The flow gradient is one-sided there and not compared.)"""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W = 2, 3, 17, 32          # W, H - 1 powers of two: (t + 0.5) * (W - 1) / W - x is exact enough to land on the node
    g = torch.Generator().manual_seed(11)
    tx = torch.randint(-3, W + 3, (N, H, W), generator=g).double()
    ty = torch.randint(-3, H + 3, (N, H, W), generator=g).double()
    flow = _flow_for_targets(tx, ty, H, W)
    img = _rand(N, C, H, W, seed=1)
    ic, ig = img.clone().requires_grad_(True), img.clone().to(device).requires_grad_(True)
    ref, got = K.flow_warp(ic, flow), ops.flow_warp(ig, flow.to(device))
    want = img[torch.arange(N).view(N, 1, 1, 1), torch.arange(C).view(1, C, 1, 1), ty.clamp(0, H - 1).long().unsqueeze(1),
               tx.clamp(0, W - 1).long().unsqueeze(1)]
    _cmp('warp_nodes.fwd_vs_gather', got, want, 1e-4)
    _cmp('warp_nodes.fwd', got, ref, 1e-4)
    wgt = _rand(*ref.shape, seed=3)
    (ref * wgt).sum().backward()
    (got * wgt.to(device)).sum().backward()
    _cmp('warp_nodes.grad_img', ig.grad, ic.grad, 1e-4)


@pytest.mark.parametrize("shape", [(2, 5, 16, 16), (3, 4, 17, 9), (1, 2, 64, 64)], ids=str)
def test_avgpool3s2(shape, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    _run_pair(ops.avgpool3s2, K.avgpool3s2, {'x': _rand(*shape, seed=1)}, device, name='avgpool')


@pytest.mark.parametrize("cfg", [((2, 6, 64, 64), 2, 2), ((2, 6, 64, 64), 32, 32), ((1, 3, 32, 32), 64, 64),
                                 ((2, 4, 48, 48), 16, 16), ((1, 2, 30, 20), 7, 9),
                                 ((2, 3, 64, 128), 16, 32), ((1, 2, 32, 16), 16, 16), ((3, 2, 24, 40), 12, 20),
                                 ((2, 3, 16, 24), 32, 48)], ids=str)      # quads: factors 4, (2, 1), 2; 2x up-sampling
def test_nearest_resize(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    shape, OH, OW = cfg
    _run_pair(ops.nearest_resize, K.nearest_resize, {'x': _rand(*shape, seed=1), 'OH': OH, 'OW': OW}, device, name='nearest')


@pytest.mark.parametrize("cfg", [((2, 6, 64, 64), [(2, 2), (4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], None),
                                 ((1, 3, 32, 64), [(32, 64), (4, 8), (16, 32)], (4, 8)),       # one level never used
                                 ((2, 5, 48, 24), [(16, 8), (48, 24), (24, 12)], None),        # factor 3, non power of two
                                 ((1, 2, 30, 20), [(7, 9), (30, 20)], None)], ids=str)         # not an integer pyramid: fallback
def test_nearest_pyramid(cfg, device):
    """ops.nearest_pyramid (fused backward over all levels) against one oracle nearest_resize per level."""
    from ag2video_amd import ops
    from oracle import kernels as K
    shape, sizes, unused = cfg
    x = _rand(*shape, seed=1)
    xc, xg = x.clone().requires_grad_(True), x.clone().to(device).requires_grad_(True)
    got = ops.nearest_pyramid(xg, sizes)
    assert len(got) == len(sizes)
    lref, lgot = 0.0, 0.0
    for i, ((h, w), yg) in enumerate(zip(sizes, got)):
        yr = K.nearest_resize(xc, h, w)
        _cmp('pyramid.fwd%d' % i, yg, yr, TOL)
        if (h, w) == unused:
            continue
        wgt = _rand(*yr.shape, seed=10 + i)
        lref = lref + (yr * wgt).sum() * (i + 1)
        lgot = lgot + (yg * wgt.to(device)).sum() * (i + 1)
    lref.backward()
    lgot.backward()
    _cmp('pyramid.grad', xg.grad, xc.grad, TOL)


def test_upsample2x(device):
    from ag2video_amd import ops
    from oracle import kernels as K
    _run_pair(ops.upsample2x, K.upsample2x, {'x': _rand(2, 5, 9, 12, seed=1)}, device, name='up2x')


@pytest.mark.parametrize("cfg", [(17, 516, 512, True, True), (136, 1152, 64, True, True), (88, 64, 4, False, True),
                                 (5, 132, 70, True, False), (200, 128, 1152, True, True), (80, 512, 512, True, True),
                                 (33, 1156, 40, False, True), (9, 130, 33, True, True), (128, 2048, 96, False, False)], ids=str)
def test_linear(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    M, Kd, N, relu, bias = cfg
    t = {'x': _rand(M, Kd, seed=1), 'w': _rand(N, Kd, seed=2, scale=Kd ** -0.5), 'b': _rand(N, seed=3) if bias else None,
         'relu': relu}
    _run_pair(ops.linear, K.linear, t, device, name='linear')


def _to_split_words(x):
    """(bf16(v) << 16) | bf16(v - bf16(v)) per value, as float32 bit patterns -- the storage format of ops.is_split tensors."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)
    return w.view(torch.float32)


@pytest.mark.parametrize("cfg", [(2, 64, 32, 32, 128, 3, 1), (3, 96, 16, 24, 384, 3, 1), (2, 40, 16, 16, 160, 3, 1), (1, 512, 8, 8, 1024, 3, 1)],
                         ids=str)
def test_conv2d_split_word_input(cfg, device):
    """A bf16x3 convolution reading its input as split words (written by the producing kernel) multiplies exactly the terms it
    derives itself from the fp32 tensor -- forward, weight gradient (plain, per frame group, shared-input form); results agree to
    the atomics-order noise of split-K launches (a dropped lo term would show at 1e-3) -- and the normalisation kernel's out_split
    output IS that format, bit for bit.  Shapes that cannot take split words are refused, not silently mis-read."""
    from ag2video_amd import ops

    def same(a, c, nm):
        tol = 1e-5 * float(a.abs().max())
        assert float((a - c).abs().max()) <= tol, (nm, float((a - c).abs().max()), tol)
    N, C, H, W, K, R, pad = cfg
    torch.manual_seed(5)
    x = torch.randn(N, C, H, W, device=device)
    x[0, 0, 0, :5] = torch.tensor([0.0, -0.0, 1e-30, 1.0, 255.5], device=device)
    w = (torch.randn(K, C, R, R, device=device) * 0.05).requires_grad_(True)
    b = torch.randn(K, device=device).requires_grad_(True)
    gs = torch.linspace(0.5, 1.5, N, device=device)
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
        assert ops.conv_split_input_ok(x.shape, w.shape, 1, pad)
        for gscale in (None, gs):
            res = []
            for split in (False, True):
                xin = (ops.mark_split(_to_split_words(x)) if split else x.clone())
                y = ops.conv2d(xin, w, b, 1, pad, gscale=gscale)
                dw, db = torch.autograd.grad((y * torch.linspace(-1, 1, y.numel(), device=device).view_as(y)).sum(), (w, b))
                res.append((y.detach(), dw, db))
            for a, c, nm in zip(res[0], res[1], ('y', 'dw', 'db')):
                same(a, c, nm)
        # shared-input form (SPADE mlp_shared): three weight sets, merged weight gradient
        ws = [(torch.randn(128, C, R, R, device=device) * 0.05).requires_grad_(True) for _ in range(3)]
        res = []
        for split in (False, True):
            xin = (ops.mark_split(_to_split_words(x)) if split else x.clone())
            ys = ops.conv2d_shared_input(xin, ws, [None] * 3, 1, pad, relu=True)
            res.append([t.detach() for t in ys] + list(torch.autograd.grad(sum((t * t).sum() for t in ys), ws)))
        for a, c in zip(res[0], res[1]):
            same(a, c, 'shared')
        # the convolution epilogue as producer (relu(mlp_shared) of a SPADE block): split words out, zeros stay zero words
        ys = ops.conv2d_shared_input(x.clone(), ws, [None] * 3, 1, pad, relu=True, grad_premasked=True, out_split=[True, False, True])
        ref = ops.conv2d_shared_input(x.clone(), ws, [None] * 3, 1, pad, relu=True, grad_premasked=True)
        assert [ops.is_split(t) for t in ys] == [True, False, True]
        for t, r, sp in zip(ys, ref, (True, False, True)):
            if sp:
                wd = t.detach().view(torch.int32)
                val = (wd & -65536).view(torch.float32) + (wd << 16).view(torch.float32)
                assert float((val - r.detach()).abs().max()) <= 3e-5 * float(r.detach().abs().max())
                assert bool(((wd == 0) == (val == 0)).all()) and bool((val >= 0).all())
            else:
                same(t.detach(), r.detach(), 'plain output')
        # the producer: batch norm + LeakyReLU with out_split
        rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
        wt, bs = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device)
        ref = ops.batchnorm_act(x, wt, bs, rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2)
        got = ops.batchnorm_act(x, wt, bs, rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2, out_split=True)
        assert ops.is_split(got) and not ops.is_split(ref)
        assert torch.equal(got.view(torch.int32), _to_split_words(ref).view(torch.int32))
        lv = ops.nearest_pyramid(got, [(H, W), (H // 2, W // 2)])
        assert all(ops.is_split(t) for t in lv) and torch.equal(lv[1].view(torch.int32), got[:, :, ::2, ::2].contiguous().view(torch.int32))
        # refused: exact arithmetic, too few rows, a weight-gradient shape without the 16-byte-load kernel
        assert not ops.conv_split_input_ok((N, C, 12, 12), w.shape, 1, pad)
        with pytest.raises(RuntimeError):       # (the forward kernels all read split words; the 12-wide weight gradient cannot)
            ops.conv2d(ops.mark_split(_to_split_words(torch.randn(N, C, 12, 12, device=device))), w, b, 1, pad).sum().backward()
    with ops.conv_precision('fp32'):
        assert not ops.conv_split_input_ok(x.shape, w.shape, 1, pad)
        with pytest.raises(RuntimeError):
            ops.conv2d(ops.mark_split(_to_split_words(x)), w, b, 1, pad)


@pytest.mark.parametrize("pair", [False, True], ids=['fused', 'pair'])
def test_spade_gradient_as_split_words(pair, device, monkeypatch):
    """gamma | beta = conv(actv) -> SPADE modulation: the modulation's backward hands d(gamma | beta) to the convolution as split
    words (data gradient, weight gradient; the bias gradient comes from the reduce pass' sums).  Same gradients as with fp32
    hand-over, to atomics noise."""
    from ag2video_amd import ops
    torch.manual_seed(11)
    N, Cn, Ch, H, W = 4, 64, 128, 32, 32
    up = 2 if pair else 1
    xs = torch.randn(N, Cn, H // up, W // up, device=device)
    actv0 = torch.relu(torch.randn(N, Ch, H, W, device=device))
    ws = [(torch.randn(2 * Cn, Ch, 3, 3, device=device) * 0.03) for _ in range(2)]
    bs = [torch.randn(2 * Cn, device=device) * 0.1 for _ in range(2)]
    wout = torch.randn(N, Cn, H, W, device=device)

    def run():
        seen = []
        actv = actv0.clone().requires_grad_(True)
        x = xs.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        bufs = [torch.zeros(Cn, device=device), torch.ones(Cn, device=device), torch.zeros(Cn, device=device), torch.ones(Cn, device=device)]
        with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), ops.frame_groups(2):
            gb0 = ops.conv2d(actv, w[0], b[0], 1, 1, relu_input=True)
            if pair:
                gb1 = ops.conv2d(actv, w[1], b[1], 1, 1, relu_input=True)
                y0, y1 = ops.spade_pair(x, gb0, gb1, bufs[0], bufs[1], bufs[2], bufs[3], 2, 0.1, 1e-5, 0.2, 1.0, 2)
                loss = (y0 * wout).sum() + 0.5 * (y1 * wout).sum()
                leaves = [actv, x, w[0], w[1], b[0], b[1]]
            else:
                y = ops.spade_norm_fused(x, gb0, bufs[0], bufs[1], True, 0.1, 1e-5, 0.2, groups=2)
                loss = (y * wout).sum()
                leaves = [actv, x, w[0], b[0]]
            gb0.register_hook(lambda g: seen.append(ops.is_split(g)))
            grads = torch.autograd.grad(loss, leaves)
        return grads, seen
    split, seen = run()
    assert seen == [True]
    monkeypatch.setenv('AG2V_NO_SPLIT_GRAD', '1')
    plain, seen = run()
    assert seen == [False]
    for a, c in zip(split, plain):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())


@pytest.mark.parametrize("cfg", [(4, 6, 32, 32, 2, [(32, 32), (16, 16), (8, 8), (2, 2)]), (3, 5, 16, 24, 1, [(8, 12), (4, 6)]),
                                 (2, 4, 8, 8, 2, [(8, 8)]),
                                 # half-size level written by the normalisation pass; three small levels folded into one before the backward
                                 (2, 4, 64, 64, 1, [(64, 64), (8, 8), (32, 32), (16, 16), (4, 4)])], ids=str)
def test_batchnorm_act_pyramid(cfg, device):
    """Affine batch norm + LeakyReLU + the nearest-neighbour pyramid of the result as one node (fused backward: the level gradients are
    gathered while the norm backward loads its dy) against the oracle's two steps; frame groups; levels without a gradient."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, G, sizes = cfg
    assert ops.batchnorm_act_pyramid_supported((N, C, H, W), sizes)
    x, wt, bs = _rand(N, C, H, W, seed=1), torch.rand(C) + 0.5, _rand(C, seed=3)
    res = []
    for mod, dev in ((K, 'cpu'), (ops, device)):
        xd, wd, bd = [t.clone().to(dev).requires_grad_(True) for t in (x, wt, bs)]
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        outs = mod.batchnorm_act_pyramid(xd, wd, bd, rm, rv, sizes, True, 0.1, 1e-5, 0.2, groups=G)
        loss = 0
        for i, o in enumerate(outs):
            if i == 1 and len(outs) > 2:
                continue                      # a level nobody back-propagates through
            loss = loss + (o * _rand(*o.shape, seed=10 + i).to(dev)).sum() * (0.5 + i)
        loss.backward()
        res.append(list(outs) + [xd.grad, wd.grad, bd.grad, rm, rv])
    for i, (got, ref) in enumerate(zip(res[1], res[0])):
        _cmp('bn_pyramid[%d]' % i, got, ref, TOL)


def test_linear_unaligned_operands(device):
    """The one-round-trip GEMM reads k-contiguous operands with 16-byte loads when their rows are 16-byte aligned and with 4-byte
    loads otherwise (views into a flat buffer at any element offset): same results either way, forward and both gradients."""
    from ag2video_amd import ops
    from oracle import kernels as K
    M, Kd, N = 80, 512, 128
    x, w, b = _rand(M, Kd, seed=1), _rand(N, Kd, seed=2, scale=Kd ** -0.5), _rand(N, seed=3)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref = K.linear(ref_in[0], ref_in[1], ref_in[2], True)
    wgt = _rand(M, N, seed=4)
    (ref * wgt).sum().backward()
    for xo, wo in ((0, 0), (1, 0), (0, 2), (3, 1)):
        xb = torch.zeros(M * Kd + 4, device=device); wb = torch.zeros(N * Kd + 4, device=device)
        xd = xb[xo:xo + M * Kd].view(M, Kd); wd = wb[wo:wo + N * Kd].view(N, Kd)
        xd.copy_(x); wd.copy_(w)
        assert xd.data_ptr() % 16 == 4 * xo and wd.data_ptr() % 16 == 4 * wo
        xd.requires_grad_(True); wd.requires_grad_(True)
        bd = b.clone().to(device).requires_grad_(True)
        got = ops.linear(xd, wd, bd, True)
        _cmp('linear_unaligned.fwd', got, ref, TOL)
        gx, gw, gb = torch.autograd.grad((got * wgt.to(device)).sum(), (xd, wd, bd))
        for nm, g, r in (('x', gx, ref_in[0].grad), ('w', gw, ref_in[1].grad), ('b', gb, ref_in[2].grad)):
            _cmp('linear_unaligned.grad_' + nm, g, r, TOL)


def _edges(B, T, O, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, O, (B, T, 2), generator=g, dtype=torch.int64)


@pytest.mark.parametrize("cfg", [(2, 6, 9, 64, 128), (3, 11, 17, 512, 128), (1, 3, 1, 40, 24)], ids=str)
def test_triple_gather(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    B, O, T, Din, Dp = cfg
    t = {'obj': _rand(B, O, Din, seed=1), 'pred': _rand(B, T, Dp, seed=2), 'edges': _edges(B, T, O, 3)}
    _run_pair(ops.triple_gather, K.triple_gather, t, device, name='gather')


@pytest.mark.parametrize("cfg", [(2, 6, 9, 64, 128), (3, 11, 17, 512, 128), (2, 4, 5, 32, 16)], ids=str)
def test_triple_pool(cfg, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    B, O, T, Hd, Dp = cfg
    valid = torch.rand(B, T, generator=torch.Generator().manual_seed(5)) > 0.3
    valid[0, :] = True
    if B > 1:
        valid[1, :] = False       # a sample whose edges are all padding: pooled stays zero
    t = {'newt': _rand(B, T, 2 * Hd + Dp, seed=1), 'edges': _edges(B, T, O, 3), 'valid': valid, 'O': O, 'Hd': Hd, 'Dp': Dp}
    _run_pair(ops.triple_pool, K.triple_pool, t, device, name='pool')


def test_embedding(device):
    from ag2video_amd import ops
    from oracle import kernels as K
    idx = torch.randint(0, 10, (3, 7), generator=torch.Generator().manual_seed(1))
    t = {'idx': idx, 'table': _rand(10, 96, seed=2)}
    _run_pair(ops.embedding, K.embedding, t, device, name='embedding')


@pytest.mark.parametrize("n", [5, 1024, 70001])
def test_activations(n, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    _run_pair(ops.leaky_relu, K.leaky_relu, {'x': _rand(n, seed=1), 'slope': 0.2}, device, name='lrelu')
    _run_pair(ops.tanh, K.tanh, {'x': _rand(n, seed=1)}, device, name='tanh')


@pytest.mark.parametrize("shape", [(7,), (2, 1, 35, 35), (3, 16, 65, 65)], ids=str)
def test_losses(shape, device):
    from ag2video_amd import ops
    from oracle import kernels as K
    a, b = _rand(*shape, seed=1), _rand(*shape, seed=2)
    _run_pair(ops.l1_mean, K.l1_mean, {'a': a, 'b': b}, device, nondiff=('b',), name='l1')
    for kind in ('g', 'd_real', 'd_fake'):
        _run_pair(ops.hinge_mean, K.hinge_mean, {'x': a * 2, 'kind': kind}, device, name='hinge_' + kind)


def test_errors_raise(device):
    """Bad arguments and CPU tensors raise RuntimeError (never abort, never fall back)."""
    from ag2video_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.randn(1, 3, 8, 8), torch.randn(4, 3, 3, 3))          # CPU tensors
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.randn(1, 3, 8, 8, device=device), torch.randn(4, 5, 3, 3, device=device))  # channel mismatch
    with pytest.raises(RuntimeError):
        ops.layout_render(torch.randn(1, 40, 8, device=device), torch.rand(1, 40, 4, device=device),
                          torch.ones(1, 40, dtype=torch.int32, device=device), 16, 16)  # > MAXO objects


def test_fused_adam_matches_torch_adam(device):
    """FusedAdam (flat buffers, one HIP kernel) against torch.optim.Adam run on the CPU: 4 steps, including a parameter
    that never receives a gradient and a state_dict round trip."""
    from ag2video_amd.optim import FusedAdam
    shapes = [(7,), (5, 3, 3, 3), (130, 17), (1,), (64, 9)]
    gen = torch.Generator().manual_seed(3)
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=gen)) for s in shapes]
    gpu_p = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2, betas=(0.5, 0.999))
    opt = FusedAdam(gpu_p, lr=1e-2, betas=(0.5, 0.999))
    for it in range(4):
        if it == 2:     # checkpoint round trip in the torch.optim.Adam layout
            sd = opt.state_dict()
            assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
            opt.load_state_dict(sd)
        ref.zero_grad()
        opt.zero_grad()
        for i, (a, b) in enumerate(zip(ref_p, gpu_p)):
            if i == 3:
                continue        # never gets a gradient
            g = torch.randn(a.shape, generator=gen)
            a.grad = g.clone()
            (b * g.to(device)).sum().backward()      # autograd accumulates into the flat view
        ref.step()
        opt.step()
    for a, b in zip(ref_p, gpu_p):
        _cmp('adam.param', b, a, 1e-5)
    assert gpu_p[1].data_ptr() >= opt.flat_param.data_ptr()
    # every parameter (and gradient) view starts on a 16-byte boundary of the flat buffers; the padding stays zero
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in gpu_p)
    used = torch.zeros_like(opt.flat_param, dtype=torch.bool)
    for off, n, _ in opt._views:
        used[off:off + n] = True
    assert float(opt.flat_param[~used].abs().sum()) == 0.0 and float(opt.flat_m[~used].abs().sum()) == 0.0


def test_gradient_sinks_multi_use_and_aborted_backward(device):
    """A weight / bias / spectral-norm weight_orig used by several calls of one graph collects its gradient in one buffer per
    backward pass (ops._wsink_*).  (i) the sums match the oracle's plain autograd; (ii) a backward pass that dies after its
    first weight-gradient launch leaves nothing behind that a later pass would accumulate into."""
    from ag2video_amd import ops
    from oracle import kernels as K
    x1, x2, x3 = _rand(2, 16, 12, 12, seed=1), _rand(2, 16, 12, 12, seed=2), _rand(2, 16, 12, 12, seed=3)
    w0, b0 = _rand(136, 16, 3, 3, seed=4, scale=0.1), _rand(136, seed=5)
    u0 = torch.nn.functional.normalize(_rand(136, seed=6), dim=0)
    v0 = torch.nn.functional.normalize(_rand(16 * 9, seed=7), dim=0)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    def run(mod, dev, boom=False):
        w = w0.clone().to(dev).requires_grad_(True)
        b = b0.clone().to(dev).requires_grad_(True)
        u, v = u0.clone().to(dev), v0.clone().to(dev)
        loss = 0.0
        for i, x in enumerate((x1, x2, x3)):
            xx = x.clone().to(dev).requires_grad_(True)
            if boom and i == 0:
                xx = Boom.apply(xx)
            weff = mod.spectral_weight(w, u, v, True, 1e-12)          # one node per use, like the 15 frame passes
            y = mod.conv2d(xx, weff, b, 1, 1)
            loss = loss + (y * _rand(*y.shape, seed=40 + i).to(dev)).sum()
        return loss, w, b

    loss, w, b = run(ops, device, boom=True)
    with pytest.raises(RuntimeError, match="boom"):
        loss.backward()
    loss, w, b = run(ops, device)
    loss.backward()
    lr, wr, br = run(K, 'cpu')
    lr.backward()
    _cmp('sink.dw_orig', w.grad, wr.grad, TOL)
    _cmp('sink.db', b.grad, br.grad, TOL)


def test_direct_param_grads_only_inside_the_accumulating_context(device):
    """a leaf with a preallocated .grad (every FusedAdam parameter) takes the direct-to-.grad path ONLY inside
    ops.accumulate_into_param_grads() (the training step's backward).  torch.autograd.grad and a plain backward() go through
    autograd: autograd.grad returns the gradient and leaves .grad alone; both routes produce the same numbers; gradients written
    from a side stream are joined to the caller's stream by the end-of-pass callback."""
    from ag2video_amd import ops
    from ag2video_amd.optim import FusedAdam
    x = _rand(2, 16, 12, 12, seed=1).to(device)
    w = torch.nn.Parameter(_rand(136, 16, 3, 3, seed=4, scale=0.1).to(device))
    b = torch.nn.Parameter(_rand(136, seed=5).to(device))
    opt = FusedAdam([w, b], lr=1e-3)
    tgt = _rand(2, 136, 12, 12, seed=9).to(device)

    def loss():
        return (ops.conv2d(x, w, b, 1, 1) * tgt).sum()
    opt.zero_grad()
    gw, gb = torch.autograd.grad(loss(), [w, b])
    assert gw is not None and gb is not None and float(gw.abs().sum()) > 0
    assert float(w.grad.abs().sum()) == 0.0 and float(b.grad.abs().sum()) == 0.0           # .grad untouched by autograd.grad
    loss().backward()                                                                        # plain backward: AccumulateGrad
    torch.cuda.synchronize()
    plain_w, plain_b = w.grad.clone(), b.grad.clone()
    _cmp('direct.autograd_grad_w', gw, plain_w.cpu(), 1e-5)
    opt.zero_grad()
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        l2 = loss()                                                                          # forward on the side stream => its backward too
    with ops.accumulate_into_param_grads():
        l2.backward()
    assert w.grad.data_ptr() == opt.flat_grad.data_ptr() or w.grad.data_ptr() > 0          # still the flat view
    got_w, got_b = w.grad.clone(), b.grad.clone()                                            # current stream: must see the side stream's writes
    torch.cuda.synchronize()
    _cmp('direct.context_w', got_w, plain_w.cpu(), 1e-5)
    _cmp('direct.context_b', got_b, plain_b.cpu(), 1e-5)
    hooked = []
    w.register_post_accumulate_grad_hook(lambda p: hooked.append(1))
    opt.zero_grad()
    with ops.accumulate_into_param_grads():
        loss().backward()
    assert hooked, "a post-accumulate-grad hook must still fire (the direct path steps aside for hooked parameters)"


# ------------------------------------------------------------------------------------------------ fp16 3-term split (forward)
@pytest.mark.parametrize("case", [(4, 256, 32, 256, 3, 1, 1), (3, 128, 40, 64, 3, 1, 1), (2, 64, 64, 32, 3, 1, 1), (3, 64, 33, 128, 3, 2, 1),
                                  (2, 48, 24, 128, 3, 1, 1)], ids=lambda c: "x".join(str(v) for v in c))
def test_conv_forward_fp16_split(case, device):
    """ag2v_conv_opts::precision = 2 (ops.bf16x3_min_rows(..., fwd_h16=True)): the forward's three products on fp16 hi / lo parts.  On
    normalised activations and kaiming-scale filters (the flow network's operands) the result must sit at the fp32 level: against an
    fp64 convolution it may not be worse than 3x the exact-fp32 kernels' own error, and must beat the bf16 split; gradients (bf16
    split in both cases) are unchanged.  Shapes: 256- / 128-row tiles, quad-load, tap-major and table (C % 32 != 0) paths, stride 2."""
    from ag2video_amd import ops
    N, C, H, Ko, R, st, pad = case
    x0 = torch.nn.functional.leaky_relu(_rand(N, C, H, H, seed=1), 0.2)
    w0 = _rand(Ko, C, R, R, seed=2, scale=(2.0 / (C * R * R)) ** 0.5)
    b0 = _rand(Ko, seed=3)
    gs0 = torch.tensor([0.7], dtype=torch.float32)
    ref = torch.nn.functional.conv2d(x0.double(), w0.double() * 0.7, b0.double(), stride=st, padding=pad)

    def run(ctx):
        x = x0.clone().to(device).requires_grad_(True)
        w = w0.clone().to(device).requires_grad_(True)
        with ops.conv_precision('bf16x3'), ctx:
            y = ops.conv2d(x, w, b0.to(device), st, pad, gscale=gs0.to(device))
            y.backward(_rand(*y.shape, seed=5).to(device))
        return y.detach().double().cpu(), x.grad.detach().cpu(), w.grad.detach().cpu()
    yh, dxh, dwh = run(ops.bf16x3_min_rows(32, 32, fwd_h16=True))
    yb, dxb, dwb = run(ops.bf16x3_min_rows(32, 32))
    ye, _, _ = run(ops.bf16x3_min_rows(ops.EXACT, 32))
    scale = ref.abs().max().item()
    eh, eb, ee = [(t - ref).abs().max().item() / scale for t in (yh, yb, ye)]
    print('fp16 split forward %s: error / max|ref|  h16 %.2e  bf16 %.2e  exact fp32 %.2e' % (case, eh, eb, ee))
    assert eh <= 3.0 * ee + 2e-7, (eh, ee)
    assert eh < 0.6 * eb, (eh, eb)
    for a, b in ((dxh, dxb), (dwh, dwb)):                     # same backward arithmetic (split-K atomics: reduction-order noise)
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


# ------------------------------------------------------------------------------------------------ deterministic reductions
DET_CASES = [
    # N, C, H, K, R, stride, pad, groups, what exercises it
    (2, 1024, 8, 1024, 3, 1, 1, 1),      # small map: forward / data gradient normally split K over workgroups (atomics)
    (6, 64, 64, 128, 3, 1, 1, 3),        # 24k pixels: the weight gradient splits its pixel range; 3 sample groups (spectral-norm form)
    (4, 128, 33, 256, 4, 2, 2, 1),       # discriminator layer: 4x4 stride 2 (four data-gradient classes), odd size
    (8, 16, 40, 24, 3, 1, 1, 1),         # 24 rows: the exact-fp32 32-row tiles (forward, data gradient, weight gradient)
    (3, 259, 24, 64, 4, 2, 2, 1),        # table-path gather (C % 32 != 0), 64-row tiles
]


@pytest.mark.parametrize("case", DET_CASES, ids=lambda c: "x".join(str(v) for v in c))
@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_conv_deterministic_mode_is_bit_reproducible(case, prec, device):
    """ops.deterministic(): no floating-point atomics -- forward / data gradient never split a tile's reduction, the weight gradient
    adds its pixel-range partials from a workspace in order (ag2v_conv_opts::deterministic).  (i) the same call twice gives the same
    bits, also when accumulating into an existing gradient and with per-group weight gradients; (ii) the results agree with the default
    (atomic) mode to reduction-order noise and with the oracle at the kernel tolerance."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, Ko, R, st, pad, groups = case
    x0 = _rand(N, C, H, H, seed=1)
    w0 = _rand(Ko, C, R, R, seed=2, scale=(C * R * R) ** -0.5)
    b0 = _rand(Ko, seed=3)
    gs0 = (torch.rand(groups, generator=torch.Generator().manual_seed(4)) + 0.5) if groups > 1 else None

    def run(mod, dev, det):
        x = x0.clone().to(dev).requires_grad_(True)
        w = w0.clone().to(dev).requires_grad_(True)
        b = b0.clone().to(dev).requires_grad_(True)
        gs = None if gs0 is None else gs0.clone().to(dev).requires_grad_(True)
        ctx = ops.deterministic(det) if mod is ops else contextlib.nullcontext()
        with ctx, (ops.bf16x3_min_rows(48) if mod is ops else contextlib.nullcontext()):
            y = mod.conv2d(x, w, b, st, pad, gscale=gs)
            dy = _rand(*y.shape, seed=5).to(dev)
            # two uses of the weight in one graph: the second weight-gradient launch ACCUMULATES into the first one's buffer
            y2 = mod.conv2d(x * 0.5, w, None, st, pad, gscale=gs)
            ((y * dy).sum() + (y2 * dy).sum() * 0.25).backward()
        return [t.detach().clone() for t in (y, x.grad, w.grad, b.grad)] + ([gs.grad.detach().clone()] if gs is not None else [])
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        a = run(ops, device, True)
        b = run(ops, device, True)
        c = run(ops, device, False)
    finally:
        ops.set_conv_precision(old)
    ref = run(K, 'cpu', False)
    names = ['y', 'dx', 'dw', 'db', 'dgscale']
    for n, ta, tb, tc, tr in zip(names, a, b, c, ref):
        assert torch.equal(ta, tb), "deterministic mode: %s differs between two runs" % n
        _cmp('det.%s vs default mode' % n, ta, tc.cpu(), 2e-5)
        _cmp('det.%s vs oracle' % n, ta, tr, TOL)


def test_spectral_bank_and_train_step_deterministic(device):
    """The remaining reductions with atomics in the default mode -- the multi-layer power iteration (W^T u over row chunks), the per-group
    scale gradients (sn_group_combine), the bias-gradient channel sums -- through one toy training iteration: with ops.deterministic()
    two runs from the same state give bit-identical outputs, losses, gradients and updated state (the engine then also keeps the graph
    nets on the main stream)."""
    from ag2video_amd import ops
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '16', '--ndf', '16', '--nff', '16',
             '--gconv_hidden_dim', '64', '--gpu_ids', '0']
    res = []
    with ops.deterministic():
        for rep in range(2):
            opt = make_opt(cater_vocab(), flags)
            ts = TrainStep(opt, device, weights_seed=5)
            assert ts.side_stream is None
            clip = ts.to_device(make_clip_batch(opt.vocab, 2, 4, (64, 64), seed=11))
            out, GL, DL = ts.gd_step(clip, apply_updates=False)
            gG, gD = ts.optimizer_generator.flat_grad.clone(), ts.discriminator.optimizer_d_img.flat_grad.clone()
            ts.gd_step(clip)
            _, GG = ts.graph_step(ts.to_device(make_clip_batch(opt.vocab, 2, 8, (64, 64), seed=12)))
            torch.cuda.synchronize()
            res.append(([t.clone() for t in out[:3]], {k: float(v) for k, v in list(GL.items()) + list(DL.items()) + list(GG.items())}, gG, gD,
                        {k: v.clone() for k, v in ts.model.state_dict().items()}))
            del ts
    (o0, l0, gG0, gD0, s0), (o1, l1, gG1, gD1, s1) = res
    assert l0 == l1, (l0, l1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(gG0, gG1) and torch.equal(gD0, gD1)
    bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    assert not bad, bad[:10]


# ------------------------------------------------------------------------------------------------ frame-group forms
# (frame-batched generator: ops.frame_groups).  The oracle restates each of them as the reference's per-pass op applied group by group.
GS_CASES = [
    # N, groups, C, H, W, K, R, stride, pad, bias, residual
    (6, 3, 16, 12, 12, 40, 3, 1, 1, True, True),       # exact tiles, residual
    (8, 4, 64, 8, 8, 136, 3, 1, 1, True, False),       # bf16x3 128 x 256 tile (tap-major, quad loads)
    (6, 2, 96, 8, 8, 256, 3, 1, 1, False, False),      # bf16x3 256 x 128 tile, 3 samples per group
    (4, 4, 1027, 8, 8, 32, 3, 1, 1, False, False),     # table path, one sample per group, split-K epilogue (atomics x scale)
    (6, 3, 3, 16, 16, 200, 3, 1, 1, False, False),     # <= 4 input channels: data gradient as 1x1 GEMM + shifted-plane sum
    (4, 2, 24, 10, 10, 3, 3, 1, 1, True, False),       # <= 4 output rows: few-row kernel
    (6, 3, 20, 9, 9, 48, 4, 2, 2, True, False),        # 4x4 stride 2: four parity classes in the data gradient
    (4, 2, 128, 6, 6, 130, 1, 1, 0, False, False),     # 1x1 (conv_s of a SPADE block)
]


@pytest.mark.parametrize("case", GS_CASES, ids=str)
def test_conv2d_group_scale(case, precision, device):
    """y_n = gscale[group(n)] * conv(x_n, w) + bias + residual: forward, dx, dw (through the per-group weight gradients and
    ag2v_sn_group_combine), dbias, dresidual and dgscale."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, G, C, H, W, Ko, R, st, pad, has_b, has_r = case
    OH, OW = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    t = {'x': _rand(N, C, H, W, seed=1), 'w': _rand(Ko, C, R, R, seed=2, scale=(C * R * R) ** -0.5),
         'bias': _rand(Ko, seed=3) if has_b else None, 'stride': st, 'pad': pad,
         'residual': _rand(N, Ko, OH, OW, seed=4) if has_r else None,
         'gscale': torch.linspace(0.6, 1.7, G) * torch.tensor([1.0, -1.0] * G)[:G]}
    with ops.bf16x3_min_rows(48):
        _run_pair(ops.conv2d, K.conv2d, t, device, name='conv_gscale')


def test_conv2d_sample_chunks(precision, device, monkeypatch):
    """Batches whose gathered operand exceeds the 4 GiB of a buffer descriptor are convolved in sample chunks inside the call
    (AG2V_CHUNK_BYTES lowers the limit): forward, data gradient, weight gradient (accumulated over chunks) and the per-group
    weight gradients must not depend on the chunking."""
    from ag2video_amd import ops
    N, G, C, H, W, Ko = 12, 6, 64, 8, 8, 136
    x0, w0 = _rand(N, C, H, W, seed=1), _rand(Ko, C, 3, 3, seed=2, scale=(C * 9) ** -0.5)
    gs = torch.linspace(0.7, 1.4, G).to(device)
    dy = _rand(N, Ko, H, W, seed=5).to(device)
    res = []
    for limit in (None, str(5 * C * H * W * 4)):       # 5 samples per chunk -> rounded down to whole groups of 2: chunks of 4
        if limit is None:
            monkeypatch.delenv('AG2V_CHUNK_BYTES', raising=False)
        else:
            monkeypatch.setenv('AG2V_CHUNK_BYTES', limit)
        out = []
        for g in (None, gs):
            x, w = x0.clone().to(device).requires_grad_(True), w0.clone().to(device).requires_grad_(True)
            gg = None if g is None else g.clone().requires_grad_(True)
            with ops.bf16x3_min_rows(48):
                y = ops.conv2d(x, w, None, 1, 1, gscale=gg)
                y.backward(dy)
            out += [y.detach(), x.grad, w.grad] + ([] if gg is None else [gg.grad])
        res.append(out)
    monkeypatch.delenv('AG2V_CHUNK_BYTES', raising=False)
    for a, b in zip(*res):
        _cmp('chunked', b, a.cpu(), 2e-5)


@pytest.mark.parametrize("cfg", [((6, 5, 9, 9), 3, 0.2), ((8, 16, 64, 64), 4, 1.0), ((4, 3, 70, 70), 2, 0.2), ((15, 8, 4, 4), 15, 0.2)], ids=str)
def test_norm_frame_groups(cfg, device):
    """Batch norm (affine) and the SPADE modulation with statistics per group of consecutive samples, and the running statistics
    after the per-group updates (in order)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    shape, G, slope = cfg
    N, C = shape[0], shape[1]
    for kind in ('bn', 'spade'):
        bufs = {}
        def mk():
            return torch.linspace(-0.5, 0.5, C), torch.linspace(0.5, 1.5, C)
        def f(mod, dev):
            rm, rv = mk()
            rm, rv = rm.to(dev), rv.to(dev)
            bufs[dev] = (rm, rv)
            if kind == 'bn':
                return lambda x, weight, bias: mod.batchnorm_act(x, weight, bias, rm, rv, True, 0.1, 1e-5, slope, groups=G)
            return lambda x, gb: mod.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, slope, groups=G)
        x = _rand(*shape, seed=1) * torch.linspace(0.5, 2.0, N).view(N, 1, 1, 1) + torch.linspace(-1, 1, N).view(N, 1, 1, 1)
        if kind == 'bn':
            t = {'x': x, 'weight': _rand(C, seed=2) + 1.0, 'bias': _rand(C, seed=3)}
        else:
            t = {'x': x, 'gb': _rand(N, 2 * C, *shape[2:], seed=2) * 0.3}
        _run_pair(f(ops, device), f(K, 'cpu'), t, device, name='norm_groups_' + kind)
        for got, ref, nm in zip(bufs[device], bufs['cpu'], ('running_mean', 'running_var')):
            _cmp('norm_groups_%s.%s' % (kind, nm), got, ref, 1e-5)


@pytest.mark.parametrize("shape", [(8, 5, 3, 3), (64, 16, 3, 3), (256, 130, 1, 1), (1024, 128, 3, 3)], ids=str)
@pytest.mark.parametrize("T", [1, 3, 15])
def test_spectral_scales(shape, T, device):
    """T consecutive power iterations: 1 / sigma_t, the advanced u / v, and the gradient of sum_t c_t / sigma_t w.r.t. weight_orig."""
    from ag2video_amd import ops
    from oracle import kernels as K
    M, R = shape[0], shape[1] * shape[2] * shape[3]
    w = _rand(*shape, seed=1)
    u0 = torch.nn.functional.normalize(_rand(M, seed=2), dim=0)
    v0 = torch.nn.functional.normalize(_rand(R, seed=3), dim=0)
    coef = _rand(T, seed=4)
    outs = []
    for mod, dev in ((K, 'cpu'), (ops, device)):
        wd = w.clone().to(dev).requires_grad_(True)
        u, v = u0.clone().to(dev), v0.clone().to(dev)
        sc = mod.spectral_scales(wd, u, v, T, True, 1e-12)
        (sc * coef.to(dev)).sum().backward()
        outs.append((sc.detach(), u, v, wd.grad))
    for a, b, nm in zip(outs[1], outs[0], ('scales', 'u', 'v', 'grad_w')):
        _cmp('spectral_scales.' + nm, a, b, TOL)
    # eval mode: one value, buffers untouched
    u, v = u0.clone().to(device), v0.clone().to(device)
    sc = ops.spectral_scales(w.to(device), u, v, T, False, 1e-12)
    ref = K.spectral_scales(w, u0.clone(), v0.clone(), T, False, 1e-12)
    assert sc.numel() == 1 and torch.equal(u.cpu(), u0) and torch.equal(v.cpu(), v0)
    _cmp('spectral_scales.eval', sc, ref, TOL)


def test_spectral_bank_matches_per_layer_scales(device):
    """ops.SpectralBank (every layer of a network per launch) against the per-layer oracle: scales, advanced u / v, weight gradients."""
    from ag2video_amd import ops
    from oracle import kernels as K

    class L(object):
        eps = 1e-12
    shapes = [(8, 5, 3, 3), (64, 16, 3, 3), (300, 7, 1, 1), (1024, 64, 3, 3), (32, 1027, 3, 3)]
    T = 5
    res = []
    for mod, dev in ((K, 'cpu'), (ops, device)):
        layers = []
        for i, shp in enumerate(shapes):
            l = L()
            M, R = shp[0], shp[1] * shp[2] * shp[3]
            l.weight_orig = _rand(*shp, seed=10 + i).to(dev).requires_grad_(True)
            l.weight_u = torch.nn.functional.normalize(_rand(M, seed=20 + i), dim=0).to(dev)
            l.weight_v = torch.nn.functional.normalize(_rand(R, seed=30 + i), dim=0).to(dev)
            layers.append(l)
        bank = mod.SpectralBank(layers)
        for rep in range(2):                      # second call: the cached descriptor table, buffers advanced further
            sc = bank.scales(T)
            tot = sum((s * _rand(T, seed=40 + i).to(dev)).sum() for i, s in enumerate(sc))
            for l in layers:
                l.weight_orig.grad = None
            tot.backward()
        res.append((layers, [s.detach().clone() for s in sc]))
    (lc, sc_c), (lg, sc_g) = res
    for i in range(len(shapes)):
        _cmp('bank.scales%d' % i, sc_g[i], sc_c[i])
        _cmp('bank.u%d' % i, lg[i].weight_u, lc[i].weight_u)
        _cmp('bank.v%d' % i, lg[i].weight_v, lc[i].weight_v)
        _cmp('bank.grad%d' % i, lg[i].weight_orig.grad, lc[i].weight_orig.grad)


def test_spade_bias_gradient_from_norm_backward(device):
    """gamma | beta convolution -> SPADE modulation: the convolution's bias gradient is taken from the channel sums the norm's
    backward reduce pass produces (no pass over the [N, 2C, H, W] gradient).  Against the oracle, with frame groups."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, Cin, C, H, G = 6, 16, 24, 12, 3
    a = _rand(N, Cin, H, H, seed=1)
    w, b = _rand(2 * C, Cin, 3, 3, seed=2, scale=0.1), _rand(2 * C, seed=3)
    x = _rand(N, C, H, H, seed=4) * 2 + 0.3
    wgt = _rand(N, C, H, H, seed=5)
    res = []
    for mod, dev in ((K, 'cpu'), (ops, device)):
        ad, wd, bd, xd = [t.clone().to(dev).requires_grad_(True) for t in (a, w, b, x)]
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gb = mod.conv2d(ad, wd, bd, 1, 1)
        y = mod.spade_norm_fused(xd, gb, rm, rv, True, 0.1, 1e-5, 0.2, groups=G)
        (y * wgt.to(dev)).sum().backward()
        res.append((bd.grad, wd.grad, ad.grad, xd.grad))
    for got, ref, nm in zip(res[1], res[0], ('dbias', 'dw', 'da', 'dx')):
        _cmp('spade_chain.' + nm, got, ref)


@pytest.mark.parametrize("cfg", [((6, 5, 8, 32), 1, 3), ((4, 3, 8, 16), 2, 2), ((6, 16, 32, 32), 2, 3), ((2, 4, 64, 128), 2, 1),
                                 ((15, 8, 4, 8), 2, 15), ((3, 2, 80, 64), 2, 1)], ids=str)
def test_spade_pair(cfg, device):
    """Two SPADE modulations of one input with shared statistics, the input read before its 2x nearest upsampling (up = 2): outputs,
    both running-stat pairs, and the gradients (dx summed over both branches and over the 2 x 2 replicas; d gamma|beta per branch)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    (N, C, Hs, Ws), up, G = cfg
    H, W = Hs * up, Ws * up
    assert ops.spade_pair_supported(H, W, up)
    xs = _rand(N, C, Hs, Ws, seed=1) * torch.linspace(0.5, 2.0, N).view(N, 1, 1, 1) + torch.linspace(-1, 1, N).view(N, 1, 1, 1)
    gb0, gb1 = _rand(N, 2 * C, H, W, seed=2) * 0.3, _rand(N, 2 * C, H, W, seed=3) * 0.3
    w0, w1 = _rand(N, C, H, W, seed=4), _rand(N, C, H, W, seed=5)
    res = []
    for mod, dev in ((K, 'cpu'), (ops, device)):
        xd, g0, g1 = [t.clone().to(dev).requires_grad_(True) for t in (xs, gb0, gb1)]
        bufs = [torch.linspace(-0.5, 0.5, C).to(dev), torch.linspace(0.5, 1.5, C).to(dev), torch.zeros(C).to(dev), torch.ones(C).to(dev)]
        y0, y1 = mod.spade_pair(xd, g0, g1, bufs[0], bufs[1], bufs[2], bufs[3], up, 0.1, 1e-5, 0.2, 1.0, G)
        ((y0 * w0.to(dev)).sum() + (y1 * w1.to(dev)).sum() * 0.7).backward()
        res.append([y0, y1, xd.grad, g0.grad, g1.grad] + bufs)
    names = ('y0', 'y1', 'dx', 'dgb0', 'dgb1', 'rm0', 'rv0', 'rm1', 'rv1')
    for nm, got, ref in zip(names, res[1], res[0]):
        _cmp('spade_pair.' + nm, got, ref, 1e-5 if nm.startswith('r') else TOL)
    assert not ops.spade_pair_supported(224, 224, 2) and ops.spade_pair_supported(224, 224, 1)


@pytest.mark.parametrize("cfg", [(4, 5, 24, 16, 16, 3, 2), (6, 4, 40, 12, 20, 3, 3), (2, 3, 16, 8, 8, 3, None),
                                 (3, 14, 32, 22, 64, 3, 3), (2, 22, 48, 10, 256, 3, None)], ids=str)      # (the last two: layout_conv_fwd3_kernel)
def test_layout_conv_img(cfg, device):
    """Layout channels (low rank) + image channels (dense, scaled per sample group) of one convolution in one launch; gradients to the
    taps U, the image, the image filter and the group scales."""
    from ag2video_amd import ops
    from oracle import kernels as K
    NB, J, Kc, H, W, R, G = cfg
    boxes = _boxes(NB, J, 21)
    boxes[0, 1] = 0.0
    keep = torch.ones(NB, J, dtype=torch.int32)
    keep[:, J - 1] = 0
    t = {'U': _rand(NB, J, Kc, R * R, seed=22), 'boxes': boxes, 'keep': keep, 'H': H, 'W': W, 'R': R, 'pad': 1,
         'img': _rand(NB, 3, H, W, seed=23), 'w_img': _rand(Kc, 3, R, R, seed=24, scale=0.2),
         'gscale': None if G is None else torch.linspace(0.6, 1.5, G)}
    assert ops.layout_conv_img_supported(H, W, R, 1, 0, 3)
    _run_pair(ops.layout_conv_img, K.layout_conv_img, t, device, nondiff=('boxes',), name='layout_conv_img')


@pytest.mark.parametrize("cfg", [(6, 14, 32, 22, 64, 3, 3), (4, 22, 48, 10, 256, 3, 2), (2, 5, 16, 256, 256, 3, 1)], ids=str)
def test_layout_conv_img_writes_the_batch_norm_statistics(cfg, device, monkeypatch):
    """conv_dim_in -> SynchronizedBatchNorm2d (generator.py:30,66,74): the layout convolution's launch also writes Chan partials of its
    output per (sample, channel, 4-row band), and the batch norm that follows combines them instead of reading the [N, K, H, W] tensor
    again.  The statistics must be those of the tensor (fp64 torch), and the normalised result / gradients / running buffers those of
    the separate statistics pass (AG2V_NO_LC_STATS)."""
    from ag2video_amd import ops
    NB, J, Kc, H, W, R, G = cfg
    boxes = _boxes(NB, J, 21).to(device)
    keep = torch.ones(NB, J, dtype=torch.int32, device=device)
    keep[:, J - 1] = 0
    U0, img0, wimg0 = _rand(NB, J, Kc, R * R, seed=22), _rand(NB, 3, H, W, seed=23) + 0.3, _rand(Kc, 3, R, R, seed=24, scale=0.2)
    gs0 = torch.linspace(0.6, 1.5, G)
    head = _rand(NB, Kc, H, W, seed=31).to(device)
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, '_LC_STATS', on)
        U, img, wimg, gs = [t.clone().to(device).requires_grad_(True) for t in (U0, img0, wimg0, gs0)]
        bw, bb = torch.linspace(0.5, 1.5, Kc, device=device).requires_grad_(True), torch.linspace(-0.2, 0.2, Kc, device=device).requires_grad_(True)
        rm, rv = torch.zeros(Kc, device=device), torch.ones(Kc, device=device)
        with ops.frame_groups(G), ops.deterministic(True):      # (no atomics: what is left between the two runs is the statistics' rounding)
            y = ops.layout_conv_img(U, boxes, keep, H, W, R, 1, img, wimg, gs)
            part = ops._bn_partials_of(y, G)
            assert (part is not None) == on
            if on:
                mean, invstd = ops._bn_stats(y, None, None, True, 0.1, 1e-5, G, part)[:2]
                yd = y.detach().double().view(G, NB // G, Kc, H * W).permute(0, 2, 1, 3).reshape(G * Kc, -1)
                m64, v64 = yd.mean(1), yd.var(1, unbiased=False)
                assert float((mean.double() - m64).abs().max()) <= 2e-6 * float(yd.abs().max())
                assert float((invstd.double() * torch.sqrt(v64 + 1e-5) - 1).abs().max()) <= 2e-6
            z = ops.batchnorm_act(y, bw, bb, rm, rv, True, 0.1, 1e-5, 0.2, groups=G)
            (z * head).sum().backward()
        res.append([z.detach(), rm, rv, U.grad, img.grad, wimg.grad, gs.grad, bw.grad, bb.grad])
    # the two runs differ by the last bits of mean / inv-std; the batch norm's backward cancels (g - m1 - xhat m2), and the scale gradient is one
    # global dot product on top: 1e-4 for the gradients (a partial that missed a row or a band would show at 1e-2)
    for i, (a, b) in enumerate(zip(*res)):
        err = float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-30)
        assert err <= (1e-5 if i < 3 else 1e-4), (i, err)


# ------------------------------------------------------------------------------------------------ lean tape (ops.set_lean_tape)
def _tape_bytes(fn):
    """Runs fn() under saved-tensor hooks; returns (result, bytes of unique device storages autograd kept for the backward)."""
    seen = {}

    def pack(t):
        if t.is_cuda:
            st = t.untyped_storage()
            seen[st.data_ptr()] = st.nbytes()
        return t
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        out = fn()
    return out, sum(seen.values())


def test_lean_tape_spade_backward_reads_compact_gamma(device):
    """ops.set_lean_tape: the SPADE modulations (pair and fused forms) keep a compact copy of gamma instead of the [N, 2C, H, W]
    gamma | beta tensor.  Same kernels, same order of operations: outputs and every gradient are bit-identical to the default
    mode's, and the tape is smaller by the beta halves."""
    from ag2video_amd import ops
    N, C, Hs, Ws, up, G = 6, 16, 8, 16, 2, 3
    H, W = Hs * up, Ws * up
    xs = _rand(N, C, Hs, Ws, seed=1)
    gb0, gb1 = _rand(N, 2 * C, H, W, seed=2) * 0.3, _rand(N, 2 * C, H, W, seed=3) * 0.3
    xf = _rand(N, C, H, W, seed=6)
    w0, w1, w2 = _rand(N, C, H, W, seed=4).to(device), _rand(N, C, H, W, seed=5).to(device), _rand(N, C, H, W, seed=7).to(device)
    res, tape = [], []
    try:
        for lean in (False, True):
            ops.set_lean_tape(lean)
            xd, g0, g1, xg, g2 = [t.clone().to(device).requires_grad_(True) for t in (xs, gb0, gb1, xf, gb0)]
            bufs = [torch.zeros(C, device=device), torch.ones(C, device=device), torch.zeros(C, device=device), torch.ones(C, device=device),
                    torch.zeros(C, device=device), torch.ones(C, device=device)]

            def fwd():
                # (g * 1.0: the modulations must not be handed leaves -- a leaf is alive whoever saves it)
                y0, y1 = ops.spade_pair(xd, g0 * 1.0, g1 * 1.0, bufs[0], bufs[1], bufs[2], bufs[3], up, 0.1, 1e-5, 0.2, 1.0, G)
                y2 = ops.spade_norm_fused(xg, g2 * 1.0, bufs[4], bufs[5], True, 0.1, 1e-5, 0.2, G)
                return y0, y1, y2
            (y0, y1, y2), nbytes = _tape_bytes(fwd)
            ((y0 * w0).sum() + (y1 * w1).sum() * 0.7 + (y2 * w2).sum()).backward()
            res.append([y0, y1, y2, xd.grad, g0.grad, g1.grad, xg.grad, g2.grad] + bufs)
            tape.append(nbytes)
    finally:
        ops.set_lean_tape(False)
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), i
    beta_bytes = 3 * N * C * H * W * 4
    assert tape[0] - tape[1] == beta_bytes, (tape, beta_bytes)


def test_lean_tape_batchnorm_recomputes_the_layout_convolution(device):
    """ops.set_lean_tape: batchnorm_act_pyramid behind layout_conv_img does not keep its [N, K, H, W] input; its backward runs the
    layout convolution's forward launch again.  Bit-identical gradients, smaller tape."""
    from ag2video_amd import ops
    NB, J, Kc, H, W, R, G = 6, 4, 40, 16, 32, 3, 3
    boxes = _boxes(NB, J, 21).to(device)
    keep = torch.ones(NB, J, dtype=torch.int32, device=device)
    keep[:, J - 1] = 0
    U0, img0, wimg0 = _rand(NB, J, Kc, R * R, seed=22), _rand(NB, 3, H, W, seed=23), _rand(Kc, 3, R, R, seed=24, scale=0.2)
    gs0 = torch.linspace(0.6, 1.5, G)
    sizes = ((H, W), (H // 2, W // 2), (H // 4, W // 4))
    wl = [_rand(NB, Kc, h, w, seed=30 + i).to(device) for i, (h, w) in enumerate(sizes)]
    res, tape = [], []
    try:
        for lean in (False, True):
            ops.set_lean_tape(lean)
            U, img, wimg, gs = [t.clone().to(device).requires_grad_(True) for t in (U0, img0, wimg0, gs0)]
            bw, bb = torch.linspace(0.5, 1.5, Kc, device=device).requires_grad_(True), torch.linspace(-0.2, 0.2, Kc, device=device).requires_grad_(True)
            rm, rv = torch.zeros(Kc, device=device), torch.ones(Kc, device=device)

            def fwd():
                y = ops.layout_conv_img(U, boxes, keep, H, W, R, 1, img, wimg, gs)
                assert ops.batchnorm_act_pyramid_supported(y.shape, sizes)
                return ops.batchnorm_act_pyramid(y, bw, bb, rm, rv, sizes, True, 0.1, 1e-5, 0.2, G)
            lv, nbytes = _tape_bytes(fwd)
            sum((l * w_).sum() for l, w_ in zip(lv, wl)).backward()
            res.append(list(lv) + [U.grad, img.grad, wimg.grad, gs.grad, bw.grad, bb.grad, rm, rv])
            tape.append(nbytes)
    finally:
        ops.set_lean_tape(False)
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), i
    assert tape[0] - tape[1] >= NB * Kc * H * W * 4 - (1 << 20), tape


@pytest.mark.parametrize("extra", [[], ['--sync_bn', '1']], ids=['default', 'sync_bn'])
def test_lean_tape_train_step_is_bit_identical(extra, precision, device):
    """One toy training iteration with --lean_tape 1 against the default tape, both with deterministic reductions: outputs, losses and
    both flat gradients are bit-identical (the lean tape recomputes or copies, it does not re-associate).  Both arithmetic modes; with
    --sync_bn 1 the modulations take the two-pass backward (ag2v_norm_bwd_reduce_g / _apply_g, mode 3)."""
    from ag2video_amd import ops
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '16', '--ndf', '16', '--nff', '16',
             '--gconv_hidden_dim', '64', '--gpu_ids', '0']
    res = []
    try:
        with ops.deterministic():
            for lean in (0, 1):
                ops.set_lean_tape(False)
                opt = make_opt(cater_vocab(), flags + extra + ['--lean_tape', str(lean)])
                ts = TrainStep(opt, device, weights_seed=5)
                assert ops.lean_tape() == bool(lean)
                clip = ts.to_device(make_clip_batch(opt.vocab, 2, 4, (64, 64), seed=11))
                out, GL, DL = ts.gd_step(clip, apply_updates=False)
                torch.cuda.synchronize()
                res.append(([t.clone() for t in out[:3]], {k: float(v) for k, v in list(GL.items()) + list(DL.items())},
                            ts.optimizer_generator.flat_grad.clone(), ts.discriminator.optimizer_d_img.flat_grad.clone()))
                del ts
    finally:
        ops.set_lean_tape(False)
        from ag2video_amd import distributed as D
        D.set_sync_bn(0)
    (o0, l0, gG0, gD0), (o1, l1, gG1, gD1) = res
    assert l0 == l1, (l0, l1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(gG0, gG1) and torch.equal(gD0, gD1)


# ------------------------------------------------------------------------------------------------ single-consumer enforcement
def test_split_word_side_channels_refuse_a_second_consumer(device):
    """The split-word hand-overs assume ONE consumer per tensor (autograd would add a second gradient contribution to split words as
    if they were floats).  The assumption is enforced: a second ops consumer of a tensor whose gradient was promised as
    split words raises in the forward pass; a second torch consumer makes the producer's backward raise; a split-word tensor
    modified in place raises at its next reader; channel sums attached to a gradient are void once it was modified."""
    from ag2video_amd import ops
    torch.manual_seed(3)
    N, Cn, Ch, H, W = 4, 64, 128, 32, 32
    actv = torch.relu(torch.randn(N, Ch, H, W, device=device)).requires_grad_(True)
    x = torch.randn(N, Cn, H, W, device=device, requires_grad=True)
    w = (torch.randn(2 * Cn, Ch, 3, 3, device=device) * 0.03).requires_grad_(True)
    b = (torch.randn(2 * Cn, device=device) * 0.1).requires_grad_(True)

    def bufs():
        return torch.zeros(Cn, device=device), torch.ones(Cn, device=device)
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), ops.frame_groups(2):
        # (1) two modulations read the same gamma | beta tensor
        gb = ops.conv2d(actv, w, b, 1, 1, relu_input=True)
        assert isinstance(getattr(gb, '_ag2v_dy_split_ok', None), ops._DySplitOffer)
        y = ops.spade_norm_fused(x, gb, *bufs(), True, 0.1, 1e-5, 0.2, groups=2)
        with pytest.raises(RuntimeError, match='second consumer'):
            ops.spade_norm_fused(x, gb, *bufs(), True, 0.1, 1e-5, 0.2, groups=2)
        y.sum().backward()          # the first consumer alone is fine
        # (2) a modulation (takes the offer) and a plain torch consumer: autograd sums split words with floats
        gb = ops.conv2d(actv, w, b, 1, 1, relu_input=True)
        y = ops.spade_norm_fused(x, gb, *bufs(), True, 0.1, 1e-5, 0.2, groups=2)
        with pytest.raises(RuntimeError, match='second consumer|modified in place'):
            (y.sum() + gb.sum()).backward()
    # (3) in-place update of a split-word tensor
    t = ops.mark_split(torch.zeros(8, device=device))
    assert ops.is_split(t)
    t.add_(1.0)
    with pytest.raises(RuntimeError, match='modified in place'):
        ops.is_split(t)
    # (4) channel sums follow the tensor's version
    g = torch.ones(2, 3, 4, 4, device=device)
    ops._attach_channel_sums(g, torch.full((3,), 32.0, device=device))
    assert ops._channel_sums_of(g) is not None
    g.mul_(2.0)
    assert ops._channel_sums_of(g) is None
