"""conv2d_shared_input's gradient slots (ops._GradSlots): the consumers of its outputs -- one conv2d(relu_input=True) each, the
gamma | beta convolution of a SPADE reading relu(mlp_shared(seg)), reference normalization.py:90-94,104-106 -- write their data
gradients into the channel slices of ONE [N, sum K_i, H, W] buffer (ag2v_conv_opts::out_nstride) that the group's merged weight /
data gradient then reads in place of torch.cat(dys).  Same kernels, same operands: every gradient must equal the torch.cat path
(AG2V_NO_GRAD_SLOTS) and an fp64 autograd evaluation of the same graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(ops, seg, ws, bs, gws, gbs, heads):
    """three mlp_shared convolutions on one map, each read by its own gamma|beta convolution, whose outputs meet fp32 heads"""
    actv = ops.conv2d_shared_input(seg, ws, bs, 1, 1, relu=True, grad_premasked=True, out_split=[True] * len(ws), grad_split=True)
    loss = 0.
    for a, gw, gb, h in zip(actv, gws, gbs, heads):
        loss = loss + (ops.conv2d(a, gw, gb, 1, 1, relu_input=True) * h).sum()
    return loss


def _ref64(seg, ws, bs, gws, gbs, heads, masks):
    """masks: where the product's forward found the pre-activation positive (a pre-activation within rounding of zero may fall on
    either side in fp64; the ReLU's subgradient there is the product's choice)"""
    F = torch.nn.functional
    loss = 0.
    for w, b, gw, gb, h, m in zip(ws, bs, gws, gbs, heads, masks):
        loss = loss + (F.conv2d(F.conv2d(seg, w, b, 1, 1) * m, gw, gb, 1, 1) * h).sum()
    return loss


@pytest.mark.parametrize("cfg", [(4, 64, 64, 64, (128, 128, 128), 96), (2, 32, 128, 128, (128, 128), 256), (3, 64, 32, 32, (128, 64, 128), 128)], ids=str)
def test_gradient_slots_equal_the_concatenation(cfg, device, monkeypatch):
    from ag2video_amd import _lib, ops
    N, C, H, W, Ks, G = cfg
    torch.manual_seed(11)
    seg0 = torch.randn(N, C, H, W, device=device)
    hi = seg0.to(torch.bfloat16)
    lo = (seg0 - hi.float()).to(torch.bfloat16)
    words = ((hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)).view(torch.float32)
    segv = hi.float() + lo.float()                     # the values the split words stand for
    ws = [torch.randn(k, C, 3, 3, device=device) * 0.06 for k in Ks]
    bs = [torch.randn(k, device=device) * 0.1 for k in Ks]
    gws = [torch.randn(G, k, 3, 3, device=device) * 0.04 for k in Ks]
    gbs = [torch.randn(G, device=device) * 0.1 for _ in Ks]
    heads = [torch.randn(N, G, H, W, device=device) for _ in Ks]
    cats = []
    real_cat = torch.cat

    def counting_cat(ts, *a, **k):
        out = real_cat(ts, *a, **k)
        cats.append(out.numel())
        return out

    def run(slots_on):
        monkeypatch.setattr(ops, '_GRAD_SLOTS', slots_on)
        monkeypatch.setattr(ops, '_SLOTS_MIN_PIXELS', 0)          # (the product keeps dense gradients below 65 536 pixels per batch: split-K)
        leaves = [t.clone().requires_grad_(True) for t in ws + bs + gws + gbs]
        n = len(Ks)
        seg = ops.mark_split(words.clone()).requires_grad_(True)
        ops.mark_split(seg)
        del cats[:]
        with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), ops.deterministic(True):
            loss = _graph(ops, seg, leaves[:n], leaves[n:2 * n], leaves[2 * n:3 * n], leaves[3 * n:], heads)
            monkeypatch.setattr(torch, 'cat', counting_cat)
            try:
                loss.backward()
            finally:
                monkeypatch.setattr(torch, 'cat', real_cat)
        return float(loss), [seg.grad] + [t.grad for t in leaves], max(cats) if cats else 0

    l1, g1, big1 = run(True)
    l0, g0, big0 = run(False)
    assert big0 >= N * sum(Ks) * H * W, "the reference path concatenates the gradients"
    assert big1 < N * min(Ks) * H * W, "with slots no activation-sized torch.cat runs in the backward"
    assert l1 == l0
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)                      # deterministic mode: the same sums in the same order
    # ... and both are the gradients of the graph (fp64 autograd on the split words' values)
    leaves = [t.double().requires_grad_(True) for t in ws + bs + gws + gbs]
    n = len(Ks)
    seg = segv.double().requires_grad_(True)
    with torch.no_grad(), ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
        actv = ops.conv2d_shared_input(ops.mark_split(words.clone()), ws, bs, 1, 1, relu=True, grad_premasked=True, out_split=[True] * n)
        masks = [((a.view(torch.int32) & 0x7FFF0000) != 0).double() * (a.view(torch.int32) >= 0).double() for a in actv]     # hi part > 0
    _ref64(seg, leaves[:n], leaves[n:2 * n], leaves[2 * n:3 * n], leaves[3 * n:], [h.double() for h in heads], masks).backward()
    refs = [seg.grad] + [t.grad for t in leaves]
    for a, r in zip(g1, refs):
        assert float((a.double() - r).abs().max()) <= 3e-4 * float(r.abs().max())


@pytest.mark.parametrize("cfg", [(4, 64, 64, 64, (128, 128, 128), 96), (2, 32, 128, 128, (128, 128), 192), (3, 64, 32, 32, (128, 64, 128), 96),
                                 (2, 32, 64, 64, (100,), 64)], ids=str)
@pytest.mark.parametrize("slots_on", [True, False], ids=['slots', 'dense'])
def test_bias_gradient_from_the_data_gradient_epilogue(cfg, slots_on, device, monkeypatch):
    """Round 6: the data gradient of conv2d(relu_input=True) -- dx = d relu(mlp_shared) -- adds dx's per-channel sums, i.e. the bias
    gradient of the convolution behind the ReLU (autograd's dy.sum((0, 2, 3)) for nn.Conv2d's bias), in the epilogue of the LDS-DMA
    kernel that writes dx (ag2v_conv_opts::out_chan_sums) instead of a pass over dx afterwards.  Both sides of ops._CHAN_SUMS_IN_DGRAD,
    with gradient slots and with dense gradients: no channel-sum launch over the activation-sized gradient on the new side, the same
    gradients (float atomics: 1e-5 of max), and the fp64 graph as reference; other kernel families and deterministic mode keep the pass."""
    import os
    from ag2video_amd import _lib, ops
    N, C, H, W, Ks, G = cfg
    torch.manual_seed(12)
    seg0 = torch.randn(N, C, H, W, device=device)
    hi = seg0.to(torch.bfloat16)
    lo = (seg0 - hi.float()).to(torch.bfloat16)
    words = ((hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)).view(torch.float32)
    segv = hi.float() + lo.float()
    n = len(Ks)
    ws = [torch.randn(k, C, 3, 3, device=device) * 0.06 for k in Ks]
    bs = [torch.randn(k, device=device) * 0.1 for k in Ks]
    gws = [torch.randn(G, k, 3, 3, device=device) * 0.04 for k in Ks]
    gbs = [torch.randn(G, device=device) * 0.1 for _ in Ks]
    heads = [torch.randn(N, G, H, W, device=device) for _ in Ks]
    big_sums = []
    real_call = _lib.call

    def counting_call(name, *a):
        if name == 'ag2v_channel_sum_acc_s' and a[2] * a[4] == N * H * W and a[3] in (tuple(Ks) + (sum(Ks),)):     # (x, out, N, C, HW, ...): dx of a gamma | beta convolution
            big_sums.append(a[3])
        return real_call(name, *a)

    def run(sums_on, det=False, halo=True):
        monkeypatch.setattr(ops, '_GRAD_SLOTS', slots_on)
        monkeypatch.setattr(ops, '_SLOTS_MIN_PIXELS', 0)
        monkeypatch.setattr(ops, '_CHAN_SUMS_IN_DGRAD', sums_on)
        old = {k: os.environ.get(k) for k in ('AG2V_HALO_MIN_TILES', 'AG2V_NO_HALO')}
        os.environ['AG2V_HALO_MIN_TILES'] = '1'
        if not halo:
            os.environ['AG2V_NO_HALO'] = '1'
        torch.cuda.synchronize()
        ops.reload_kernel_switches()
        try:
            leaves = [t.clone().requires_grad_(True) for t in ws + bs + gws + gbs]
            seg = ops.mark_split(words.clone()).requires_grad_(True)
            ops.mark_split(seg)
            del big_sums[:]
            with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), ops.deterministic(det):
                loss = _graph(ops, seg, leaves[:n], leaves[n:2 * n], leaves[2 * n:3 * n], leaves[3 * n:], heads)
                monkeypatch.setattr(_lib, 'call', counting_call)
                try:
                    loss.backward()
                finally:
                    monkeypatch.setattr(_lib, 'call', real_call)
            torch.cuda.synchronize()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            ops.reload_kernel_switches()
        return [seg.grad] + [t.grad for t in leaves], list(big_sums)

    g1, sums1 = run(True)
    g0, sums0 = run(False)
    assert not sums1, "no channel-sum pass over an activation-sized gradient: %s" % (sums1,)
    assert sums0, "the other side sums dx in its own pass"
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    _, sums_det = run(True, det=True)
    assert sums_det, "deterministic mode keeps the ordered pass"
    g_nh, sums_nh = run(True, halo=False)
    assert sums_nh, "kernels without the epilogue sums: the caller's pass"
    for a, b in zip(g_nh, g0):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    leaves = [t.double().requires_grad_(True) for t in ws + bs + gws + gbs]
    seg = segv.double().requires_grad_(True)
    with torch.no_grad(), ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
        actv = ops.conv2d_shared_input(ops.mark_split(words.clone()), ws, bs, 1, 1, relu=True, grad_premasked=True, out_split=[True] * n)
        masks = [((a.view(torch.int32) & 0x7FFF0000) != 0).double() * (a.view(torch.int32) >= 0).double() for a in actv]
    _ref64(seg, leaves[:n], leaves[n:2 * n], leaves[2 * n:3 * n], leaves[3 * n:], [h.double() for h in heads], masks).backward()
    refs = [seg.grad] + [t.grad for t in leaves]
    for a, r in zip(g1, refs):
        assert float((a.double() - r).abs().max()) <= 3e-4 * float(r.abs().max())


def test_strided_data_gradient_is_refused_where_no_kernel_writes_it(device):
    """out_nstride with a strided / accumulating / few-channel data gradient: AG2V_EINVAL, never a silently dense write."""
    from ag2video_amd import ops
    dy = torch.randn(2, 64, 16, 16, device=device)
    w = torch.randn(64, 32, 4, 4, device=device)
    buf = torch.zeros(2, 96, 32, 32, device=device)
    with pytest.raises(RuntimeError):
        ops._conv_dgrad_launch(32, dy, w, None, buf[:, 16:48], 2, 1, 0)          # stride 2
    w3 = torch.randn(64, 32, 3, 3, device=device)
    buf = torch.zeros(2, 96, 16, 16, device=device)
    with pytest.raises(RuntimeError):
        ops._conv_dgrad_launch(32, dy, w3, None, buf[:, 16:48], 1, 1, 1)         # accumulating
    with pytest.raises(RuntimeError):
        ops._conv_dgrad_launch(32, dy, w3, None, buf[:, :, ::2], 1, 1, 0)        # not a channel slice
    assert float(buf.abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [(4, 128, 64, 64, 64, 3, 1), (6, 64, 32, 64, 32, 3, 2), (2, 128, 32, 32, 48, 1, 1)], ids=str)
def test_weight_gradient_with_swapped_operands(cfg, device):
    """Layers with fewer output than input channels (up_3's 128 -> 64 convolutions, spade_generator.py:29-34; the flow network's
    64 -> 32 layers) run their same-size stride-1 weight gradient with x as the dense row operand and dy as the gathered one
    (ops._wgrad_swap_pays / _conv_wgrad_swapped: less tile padding, less producer work): against the plain launch and fp64, fp32 and
    split-word operands, per-frame-group gradients (spectral norm) and accumulation."""
    from ag2video_amd import ops
    N, C, H, W, K, R, G = cfg
    pad = (R - 1) // 2
    torch.manual_seed(5)
    x = torch.randn(N, C, H, W, device=device)
    dy = torch.randn(N, K, H, W, device=device)

    def words(t):
        hi = t.to(torch.bfloat16)
        lo = (t - hi.float()).to(torch.bfloat16)
        return ((hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)).view(torch.float32), hi.float() + lo.float()
    xw, xv = words(x)
    dw_, dv = words(dy)
    w0 = torch.zeros(K, C, R, R, device=device, dtype=torch.float64, requires_grad=True)

    def ref(xs, dys):
        y = torch.nn.functional.conv2d(xs.double(), w0, None, 1, pad)
        return torch.autograd.grad((y * dys.double()).sum(), w0)[0]
    with ops.conv_precision('bf16x3'), torch.no_grad():
        assert ops._wgrad_swap_pays(32, x.shape, dy.shape, K, R, R, 1, pad, False, False) == (R == 3)
        assert not ops._wgrad_swap_pays(32, x.shape, dy.shape, K, R, R, 2, pad, False, False)
        assert not ops._wgrad_swap_pays(32, dy.shape, x.shape, C, R, R, 1, pad, False, False)       # more output than input channels
        for xs, ds in ((False, False), (True, False), (False, True), (True, True)):
            xa, da = (xw if xs else x), (dw_ if ds else dy)
            with torch.enable_grad():
                r = ref(xv if xs else x, dv if ds else dy)
            sc = float(r.abs().max())
            plain = torch.empty(K, C, R, R, device=device)
            ops._conv_wgrad_launch(32, da, xa, plain, K, R, R, 1, pad, 0, in_split=xs, dy_split=ds)
            sw = torch.empty(K, C, R, R, device=device)
            ops._conv_wgrad_swapped(32, da, xa, sw, K, R, R, pad, 0, 1, xs, ds)
            assert float((sw.double() - r).abs().max()) <= 2e-4 * sc, (xs, ds)
            assert float((sw - plain).abs().max()) <= 2e-5 * sc, (xs, ds)
            ops._conv_wgrad_swapped(32, da, xa, sw, K, R, R, pad, 1, 1, xs, ds)                       # accumulate
            assert float((sw.double() - 2 * r).abs().max()) <= 4e-4 * sc
        if G > 1:
            g = torch.empty(G, K, C, R, R, device=device)
            ops._conv_wgrad_swapped(32, dy, x, g, K, R, R, pad, 0, G, False, False)
            for i in range(G):
                sl = slice(i * N // G, (i + 1) * N // G)
                with torch.enable_grad():
                    rg = ref(x[sl], dy[sl])
                assert float((g[i].double() - rg).abs().max()) <= 2e-4 * float(rg.abs().max())


@pytest.mark.parametrize("cfg", [(6, 40, 18, 18, 1, 4, 2), (5, 24, 20, 28, 3, 3, 1), (3, 16, 9, 33, 2, 1, 0), (120, 64, 34, 34, 1, 4, 2), (4, 8, 12, 12, 4, 4, 1)], ids=str)
def test_few_row_weight_gradient_kernel(cfg, device):
    """conv_wgrad_fewrow_kernel: <= 4 output channels (the discriminators' last layer, 512 -> 1 4x4 pad 2, discriminator.py:368-379):
    one workgroup per input channel, register-resident windows -- against fp64, accumulating, and bit-reproducible in deterministic mode."""
    from ag2video_amd import _lib, ops
    N, C, H, W, K, R, pad = cfg
    torch.manual_seed(9)
    x = torch.randn(N, C, H, W, device=device)
    OH, OW = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = torch.randn(N, K, OH, OW, device=device)
    w0 = torch.zeros(K, C, R, R, device=device, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x.double(), w0, None, 1, pad)
    ref = torch.autograd.grad((y * dy.double()).sum(), w0)[0]
    sc = float(ref.abs().max())
    with torch.no_grad():
        dw = torch.empty(K, C, R, R, device=device)
        ops._conv_wgrad_launch(None, dy, x, dw, K, R, R, 1, pad, 0)
        assert _lib.query('ag2v_last_conv_kernel') == b'conv_wgrad_fewrow_kernel', _lib.query('ag2v_last_conv_kernel')
        assert float((dw.double() - ref).abs().max()) <= 2e-5 * sc
        ops._conv_wgrad_launch(None, dy, x, dw, K, R, R, 1, pad, 1)
        assert float((dw.double() - 2 * ref).abs().max()) <= 4e-5 * sc
        with ops.deterministic(True):
            a, b = torch.empty_like(dw), torch.empty_like(dw)
            ops._conv_wgrad_launch(None, dy, x, a, K, R, R, 1, pad, 0)
            ops._conv_wgrad_launch(None, dy, x, b, K, R, R, 1, pad, 0)
            assert torch.equal(a, b)
            assert float((a.double() - ref).abs().max()) <= 2e-5 * sc
            ops._conv_wgrad_launch(None, dy, x, a, K, R, R, 1, pad, 1)
            assert float((a.double() - 2 * ref).abs().max()) <= 4e-5 * sc
        if N % 3 == 0:          # one gradient per group of N / 3 samples (spectral norm's frame groups)
            g = torch.empty(3, K, C, R, R, device=device)
            ops._conv_wgrad_launch(None, dy, x, g, K, R, R, 1, pad, 0, groups=3)
            assert _lib.query('ag2v_last_conv_kernel') == b'conv_wgrad_fewrow_kernel'
            for i in range(3):
                sl = slice(i * N // 3, (i + 1) * N // 3)
                with torch.enable_grad():
                    yg = torch.nn.functional.conv2d(x[sl].double(), w0, None, 1, pad)
                    rg = torch.autograd.grad((yg * dy[sl].double()).sum(), w0)[0]
                assert float((g[i].double() - rg).abs().max()) <= 2e-5 * float(rg.abs().max())


@pytest.mark.parametrize("cfg", [(3, 20, 34, 34, 1, 4, 2), (2, 16, 64, 64, 3, 3, 1), (5, 9, 17, 23, 2, 3, 1), (2, 12, 31, 30, 4, 4, 2), (3, 7, 10, 70, 1, 1, 0),
                                 (2, 8, 33, 258, 3, 3, 1), (2, 6, 12, 9, 2, 4, 1)], ids=str)
def test_few_row_forward_second_form(cfg, device):
    """conv_fewrow2_kernel (<= 4 output channels: conv_img spade_generator.py:28,82, conv_flow flows_generator.py:56, the
    discriminators' last layer): one 16-byte quad per input row and lane, neighbours' columns by lane exchange, two output rows per lane --
    against the oracle (oracle/kernels.py::conv2d), with bias, residual, ReLU and per-group scale; map widths that are
    not multiples of 4, waves that span several rows / images, output wider than the input (4 x 4, pad 2)."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, M, R, pad = cfg
    torch.manual_seed(13)
    x = torch.randn(N, C, H, W, device=device)
    w = torch.randn(M, C, R, R, device=device) * 0.1
    b = torch.randn(M, device=device)
    OH, OW = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    res = torch.randn(N, M, OH, OW, device=device)
    gs = torch.linspace(0.5, 1.5, N, device=device)
    for kw in (dict(bias=b), dict(bias=None, residual=res, relu=True), dict(bias=b, gscale=gs)):
        with torch.no_grad():
            out = ops.conv2d(x, w, kw.get('bias'), 1, pad, residual=kw.get('residual'), relu=kw.get('relu', False), gscale=kw.get('gscale'))
        assert _lib.query('ag2v_last_conv_kernel') == b'conv_fewrow2_kernel', _lib.query('ag2v_last_conv_kernel')
        ref = K.conv2d(x.cpu(), w.cpu(), None if kw.get('bias') is None else b.cpu(), 1, pad,
                       residual=None if kw.get('residual') is None else res.cpu(), relu=kw.get('relu', False),
                       gscale=None if kw.get('gscale') is None else gs.cpu())
        assert float((out.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), sorted(kw)


@pytest.mark.parametrize("cfg", [(3, 3, 32, 32, 16, 4, 2, 2), (2, 2, 30, 26, 8, 4, 2, 1), (4, 3, 64, 64, 64, 4, 2, 2), (2, 4, 17, 21, 8, 4, 2, 2), (2, 3, 20, 20, 8, 3, 2, 1)], ids=str)
def test_strided_data_gradient_to_few_channels(cfg, device):
    """The data gradient of a 4 x 4 stride-2 convolution with <= 4 input channels (the discriminators' first layer on the generated
    image, discriminator.py:368-379): its four parity classes are 2 x 2 stride-1 convolutions of dy that write every second pixel of
    dx -- conv_fewrow2_kernel with output placement (was: a 3-row GEMM per class on 32-row tiles, 5.7 TF).  Against fp64
    conv_transpose2d, plain, masked and accumulating; odd map sizes; a 3 x 3 filter (non-square classes keep the general kernels)."""
    from ag2video_amd import _lib, ops
    N, C, H, W, K, R, stride, pad = cfg
    torch.manual_seed(17)
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(N, K, OH, OW, device=device)
    w = torch.randn(K, C, R, R, device=device) * 0.1
    xin = torch.randn(N, C, H, W, device=device)
    ref = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), None, stride, pad,
                                               output_padding=(H + 2 * pad - R - (OH - 1) * stride, W + 2 * pad - R - (OW - 1) * stride)).float()
    sc = float(ref.abs().max())
    with torch.no_grad():
        dx = torch.empty(N, C, H, W, device=device)
        ops._conv_dgrad_launch(None, dy, w, None, dx, stride, pad, 0)
        if R == 4:
            assert _lib.query('ag2v_last_conv_kernel') == b'conv_fewrow2_kernel', _lib.query('ag2v_last_conv_kernel')
        assert float((dx - ref).abs().max()) <= 2e-5 * sc
        ops._conv_dgrad_launch(None, dy, w, xin, dx, stride, pad, 0)                     # ReLU mask of the convolution's input
        assert float((dx - ref * (xin > 0).float()).abs().max()) <= 2e-5 * sc
        base = torch.randn(N, C, H, W, device=device)
        dx = base.clone()
        ops._conv_dgrad_launch(None, dy, w, None, dx, stride, pad, 1)                    # dx += ...
        assert float((dx - base - ref).abs().max()) <= 2e-5 * sc + 1e-6


@pytest.mark.parametrize("cfg", [(3, 64, 64, 64, 3), (2, 32, 48, 72, 2), (5, 16, 20, 36, 4)], ids=str)
def test_data_gradient_of_few_row_layers_on_the_dense_image_path(cfg, device):
    """conv2d backward of a 3 x 3 layer with <= 4 output channels (conv_img spade_generator.py:28,82; conv_flow flows_generator.py:56):
    dx = same-size convolution of the 3- / 2-channel dy with the flipped, transposed filter on ag2v_layout_conv_img_fwd's image path (J = 0).
    All gradients of the node against fp64 autograd."""
    from ag2video_amd import ops
    N, C, H, W, K = cfg
    torch.manual_seed(19)
    x = torch.randn(N, C, H, W, device=device, requires_grad=True)
    w = (torch.randn(K, C, 3, 3, device=device) * 0.1).requires_grad_(True)
    b = torch.randn(K, device=device, requires_grad=True)
    head = torch.randn(N, K, H, W, device=device)
    assert ops._FEWIN_DGRAD
    (ops.conv2d(x, w, b, 1, 1) * head).sum().backward()
    xd, wd, bd = [t.detach().double().requires_grad_(True) for t in (x, w, b)]
    (torch.nn.functional.conv2d(xd, wd, bd, 1, 1) * head.double()).sum().backward()
    for got, ref in ((x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert float((got.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize("cfg", [(3, 64, 33, 33, 128, 4, 2), (2, 32, 64, 64, 64, 3, 1), (2, 96, 31, 45, 96, 4, 2), (4, 128, 16, 16, 256, 4, 2), (2, 64, 18, 18, 64, 4, 1)], ids=str)
def test_stride_2_data_gradient_as_one_convolution(cfg, device):
    """Stride-2 data gradients whose four pixel classes read the same 2 x 2 window of dy (4 x 4 pad 2: the discriminators' layers,
    discriminator.py:368-379; 3 x 3 pad 1: the flow network's down path, flows_generator.py:45-47) run as ONE stride-1 2 x 2 convolution
    to 4 C class-major channels + a depth-to-space pass (conv_igemm.hip: s2_unified_ok).  Against fp64 conv_transpose2d: plain, with the
    ReLU mask, accumulating, with per-group scales; odd map sizes (class grids of different sizes); 4 x 4 pad 1 keeps the per-class launches."""
    from ag2video_amd import _lib, ops
    N, C, H, W, K, R, pad = cfg
    torch.manual_seed(23)
    OH, OW = (H + 2 * pad - R) // 2 + 1, (W + 2 * pad - R) // 2 + 1
    dy = torch.randn(N, K, OH, OW, device=device)
    w = torch.randn(K, C, R, R, device=device) * 0.05
    xin = torch.randn(N, C, H, W, device=device)
    gs = torch.linspace(0.5, 1.5, N, device=device)
    ref = torch.nn.functional.conv_transpose2d(dy.double(), w.double(), None, 2, pad,
                                               output_padding=(H + 2 * pad - R - (OH - 1) * 2, W + 2 * pad - R - (OW - 1) * 2)).float()
    sc = float(ref.abs().max())
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
        dx = torch.empty(N, C, H, W, device=device)
        ops._conv_dgrad_launch(32, dy, w, None, dx, 2, pad, 0)
        assert float((dx - ref).abs().max()) <= 2e-4 * sc
        ops._conv_dgrad_launch(32, dy, w, xin, dx, 2, pad, 0)
        assert float((dx - ref * (xin > 0).float()).abs().max()) <= 2e-4 * sc
        base = torch.randn(N, C, H, W, device=device)
        dx = base.clone()
        ops._conv_dgrad_launch(32, dy, w, None, dx, 2, pad, 1)
        assert float((dx - base - ref).abs().max()) <= 2e-4 * sc + 1e-6
        dx = torch.empty(N, C, H, W, device=device)
        ops._conv_dgrad_launch(32, dy, w, None, dx, 2, pad, 0, gscale=gs, group_n=1)
        assert float((dx - ref * gs.view(N, 1, 1, 1)).abs().max()) <= 3e-4 * sc


@pytest.mark.parametrize("cfg", [(6, 128, 33, 33, 256, 4, 2), (4, 64, 17, 35, 128, 3, 1), (30, 256, 17, 17, 512, 4, 2)], ids=str)
def test_odd_width_weight_gradient_through_zero_columns(cfg, device):
    """conv2d backward of a stride-1 layer whose output rows are not whole runs of 8 pixels (the discriminators' 4 x 4 pad 2 layers on
    33 x 33 / 17 x 17 maps, discriminator.py:368-379): ops._wgrad_pad_pays pads dy and x by the same zero columns so that the
    16-byte-load weight-gradient kernel takes the launch.  Against fp64 autograd and the unpadded launch."""
    from ag2video_amd import _lib, ops
    N, C, H, W, K, R, pad = cfg
    torch.manual_seed(31)
    x = torch.randn(N, C, H, W, device=device)
    OH, OW = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = torch.randn(N, K, OH, OW, device=device)
    w0 = torch.zeros(K, C, R, R, device=device, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x.double(), w0, None, 1, pad)
    ref = torch.autograd.grad((y * dy.double()).sum(), w0)[0]
    sc = float(ref.abs().max())
    with ops.conv_precision('bf16x3'), torch.no_grad():
        pays = ops._wgrad_pad_pays(48, x.shape, dy.shape, K, R, R, 1, pad, False, False)
        assert pays == (OW >= 16 and N * OH * OW >= 16384 and (-OW % 8) * 5 <= OW)
        plain = torch.empty(K, C, R, R, device=device)
        ops._conv_wgrad_launch(48, dy, x, plain, K, R, R, 1, pad, 0)
        extra = -OW % 8
        padded = torch.empty(K, C, R, R, device=device)
        ops._conv_wgrad_launch(48, torch.nn.functional.pad(dy, (0, extra)), torch.nn.functional.pad(x, (0, extra)), padded, K, R, R, 1, pad, 0)
        assert b'_v_kernel' in _lib.query('ag2v_last_conv_kernel') or b'_vps_kernel' in _lib.query('ag2v_last_conv_kernel')
        assert float((padded.double() - ref).abs().max()) <= 2e-4 * sc
        assert float((padded - plain).abs().max()) <= 2e-5 * sc
