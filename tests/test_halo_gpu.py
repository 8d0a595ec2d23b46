"""The halo-tile LDS-DMA convolution kernels (csrc/conv_halo.hip: 3x3 stride-1 forward / data gradient on split-word operands) against
the oracle (oracle/kernels.py::conv2d = nn.Conv2d.forward, reference architecture.py:29-32 / normalization.py:90-94 call sites) on the
split words' exact values, and against the wave-specialised kernels they replace (AG2V_NO_HALO=1: same products, another summation order):
every patch geometry (TW x TH = 128x2, 64x4, 32x8 for the 64- / 128-row tiles; 128x1, 64x2, 32x4 for the 256-row tile), row counts
that do not fill the last row tile, every epilogue (bias, ReLU, residual, per-group scale, ReLU mask, split-word output), the data
gradient, image borders on all four sides and several images per launch.  AG2V_HALO_MIN_TILES=1 lets these small shapes take the
kernels (the product only sends launches with >= ag2v_halo_min_tiles() = 384 tiles there).  The switches are cached by the library:
ag2v_halo_reload_env() after every change."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _to_split_words(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)
    return w.view(torch.float32)


def _from_split_words(t):
    wd = t.view(torch.int32)
    return (wd & -65536).view(torch.float32) + (wd << 16).view(torch.float32)


def _reload():
    from ag2video_amd import ops
    torch.cuda.synchronize()
    ops.reload_kernel_switches()


class _halo(object):
    """on: False = the previous kernels; True / '16' = the 16x16x32 halo kernels (the default); '32' = the 32x32x16 halo kernels;
    'narrow' = the default without the 512-pixel tile of the 64-row layers (AG2V_NO_HALO_WIDE: 32 pixels per wave, as before round 6)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ('AG2V_NO_HALO', 'AG2V_HALO_MIN_TILES', 'AG2V_HALO_MFMA32', 'AG2V_NO_HALO_WIDE')}
        os.environ['AG2V_HALO_MIN_TILES'] = '1'
        os.environ.pop('AG2V_HALO_MFMA32', None)
        os.environ.pop('AG2V_NO_HALO_WIDE', None)
        if self.on:
            os.environ.pop('AG2V_NO_HALO', None)
            if self.on == '32':
                os.environ['AG2V_HALO_MFMA32'] = '1'
            if self.on == 'narrow':
                os.environ['AG2V_NO_HALO_WIDE'] = '1'
        else:
            os.environ['AG2V_NO_HALO'] = '1'
        _reload()

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _reload()
        return False


# (N, C, H, W, K): geometry it selects
CASES = [(2, 32, 32, 32, 128), (2, 64, 64, 64, 128), (1, 32, 128, 128, 128), (3, 32, 256, 256, 96),        # 128-row tile: 32x8, 64x4, 128x2 (x2 patches per row)
         (2, 32, 32, 32, 256), (2, 32, 64, 64, 320), (1, 64, 128, 128, 256),                                  # 256-row tile: 32x4, 64x2, 128x1
         (2, 32, 64, 64, 64), (1, 32, 128, 128, 48), (2, 64, 32, 32, 64),                                     # 64-row tile
         (1, 32, 224, 224, 128), (2, 96, 64, 64, 384),                                                        # c5's 224 = 7 * 32; three 128-row tiles
         (1, 32, 64, 64, 100), (1, 32, 64, 64, 40), (1, 64, 32, 32, 200)]                                     # ragged row counts: masked rows, zero-padded LDS bias


@pytest.mark.parametrize("cfg", CASES, ids=str)
def test_halo_forward_matches_oracle_and_previous_kernels(cfg, device):
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, Kout = cfg
    torch.manual_seed(3)
    x = torch.randn(N, C, H, W, device=device)
    xs = _to_split_words(x)
    xv = _from_split_words(xs)                      # the values the kernels multiply (exact in fp32)
    w = torch.randn(Kout, C, 3, 3, device=device) * 0.05
    b = torch.randn(Kout, device=device)
    res = torch.randn(N, Kout, H, W, device=device)
    gs = torch.linspace(0.5, 1.5, N, device=device)
    variants = [dict(bias=b, relu=True), dict(bias=None, residual=res), dict(bias=b, gscale=gs)]
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
        for kw in variants:
            outs = {}
            for on in (True, '32', False):
                with _halo(on):
                    outs[on] = ops.conv2d(ops.mark_split(xs.clone()), w, kw.get('bias'), 1, 1, residual=kw.get('residual'), relu=kw.get('relu', False),
                                          gscale=kw.get('gscale'))
                    name = _lib.query('ag2v_last_conv_kernel')
                    assert name.startswith({True: b'conv_halo16_bf16x3_kernel', '32': b'conv_halo_bf16x3_kernel', False: b'conv_igemm'}[on]), name
            ref = K.conv2d(xv.cpu(), w.cpu(), None if kw.get('bias') is None else b.cpu(), 1, 1,
                           residual=None if kw.get('residual') is None else res.cpu(), relu=kw.get('relu', False),
                           gscale=None if kw.get('gscale') is None else gs.cpu())
            scale = float(ref.abs().max())
            for fam in (True, '32'):
                assert float((outs[fam].cpu() - ref).abs().max()) <= 2e-4 * scale, (sorted(kw), fam, 'oracle')
                assert float((outs[fam] - outs[False]).abs().max()) <= 1e-5 * scale, (sorted(kw), fam, 'previous kernels')
        # fp32 gathered tensor (not split words): the same kernels with the 3-term split done by the multiplying wave
        for on in (True, '32', False):
            with _halo(on):
                outs[on] = ops.conv2d(x, w, b, 1, 1)
                name = _lib.query('ag2v_last_conv_kernel')
                assert (name.startswith(b'conv_halo') and (name.endswith(b'false>') or name.endswith(b'false, 4>'))) == bool(on), name
        ref = K.conv2d(x.cpu(), w.cpu(), b.cpu(), 1, 1)
        for fam in (True, '32'):
            assert float((outs[fam].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
            assert float((outs[fam] - outs[False]).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("cfg", [(2, 64, 64, 64, 128), (1, 32, 128, 128, 256), (2, 32, 64, 64, 64)], ids=str)
def test_halo_split_word_output_and_data_gradient(cfg, device):
    """out_split (relu(mlp_shared) written as split words by the epilogue) and the stride-1 data gradient with its gathered operand dy
    as split words and the ReLU mask of the convolution's input in the epilogue (ops._conv_dgrad_launch as conv2d's backward calls it)."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, Kout = cfg
    torch.manual_seed(4)
    xs = _to_split_words(torch.randn(N, C, H, W, device=device))
    w = torch.randn(Kout, C, 3, 3, device=device) * 0.05
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
        ref = K.conv2d(_from_split_words(xs).cpu(), w.cpu(), None, 1, 1, relu=True)
        for fam in (True, '32'):
            with _halo(fam):
                ys = ops.conv2d_shared_input(ops.mark_split(xs.clone()), [w], [None], 1, 1, relu=True, grad_premasked=True, out_split=[True])[0]
                assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_halo')
            got = _from_split_words(ys.detach()).cpu()
            assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
            # where the ReLU clipped, the split word is the zero word (another summation order may leave a sum within rounding of zero
            # on the other side: those stay tiny)
            assert float(got[ref == 0].abs().max()) <= 1e-5 * float(ref.abs().max())
            assert float((ys.detach().view(torch.int32)[got.to(device) == 0] & 0x7FFF7FFF).abs().max()) == 0
        # data gradient: dx = conv_transpose(dy, w) masked by (x_in > 0); dy arrives as split words
        dy = torch.randn(N, Kout, H, W, device=device)
        dys = _to_split_words(dy)
        xin = torch.randn(N, C, H, W, device=device)              # the (ReLU output) input of the forward convolution: the mask
        outs = {}
        for on in (True, '32', False):
            with _halo(on):
                dx = torch.empty(N, C, H, W, device=device)
                ops._conv_dgrad_launch(32, dys, w, xin, dx, 1, 1, 0, dy_split=True)
                assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_halo') == bool(on)
                outs[on] = dx
        dyv = _from_split_words(dys).cpu().double()
        refdx = torch.nn.functional.conv_transpose2d(dyv, w.cpu().double(), None, 1, 1).float() * (xin.cpu() > 0).float()
        sc = float(refdx.abs().max())
        for fam in (True, '32'):
            assert float((outs[fam].cpu() - refdx).abs().max()) <= 2e-4 * sc
            assert float((outs[fam] - outs[False]).abs().max()) <= 1e-5 * sc


@pytest.mark.parametrize("cfg", [(2, 64, 32, 32, 128), (3, 40, 24, 48, 256), (4, 32, 16, 16, 160)], ids=str)
def test_exact_fp32_weight_gradient_on_the_wave_specialised_pipeline(cfg, device):
    """--precision fp32: weight gradients with >= 128 rows, stride 1 and pixel rows of whole 8-pixel runs run on conv_wgrad_f32_v_kernel
    (16-byte loads, 8 producer + 8 consumer waves, v_mfma_f32_32x32x2_f32) instead of the 4-wave conv_wgrad_kernel: against an fp64
    evaluation of the same sums -- plain, accumulating, and one gradient per sample group (spectral norm's frame groups)."""
    from ag2video_amd import _lib, ops
    N, C, H, W, K = cfg
    torch.manual_seed(7)
    x = torch.randn(N, C, H, W, device=device)
    dy = torch.randn(N, K, H, W, device=device)
    w0 = torch.zeros(K, C, 3, 3, device=device, dtype=torch.float64, requires_grad=True)

    def ref(xs, dys):
        y = torch.nn.functional.conv2d(xs.double(), w0, None, 1, 1)
        return torch.autograd.grad((y * dys.double()).sum(), w0)[0].float()
    with ops.conv_precision('fp32'), torch.no_grad():
        dw = torch.empty(K, C, 3, 3, device=device)
        ops._conv_wgrad_launch(None, dy, x, dw, K, 3, 3, 1, 1, 0)
        assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_wgrad_f32_v_kernel'), _lib.query('ag2v_last_conv_kernel')
        with torch.enable_grad():
            r = ref(x, dy)
        sc = float(r.abs().max())
        assert float((dw - r).abs().max()) <= 2e-5 * sc
        ops._conv_wgrad_launch(None, dy, x, dw, K, 3, 3, 1, 1, 1)          # accumulate: dw += the same gradient
        assert float((dw - 2 * r).abs().max()) <= 4e-5 * sc
        if N % 2 == 0:
            G = 2
            dwg = torch.empty(G, K, C, 3, 3, device=device)
            ops._conv_wgrad_launch(None, dy, x, dwg, K, 3, 3, 1, 1, 0, groups=G)
            for g in range(G):
                sl = slice(g * N // G, (g + 1) * N // G)
                with torch.enable_grad():
                    rg = ref(x[sl], dy[sl])
                assert float((dwg[g] - rg).abs().max()) <= 2e-5 * float(rg.abs().max())


@pytest.mark.parametrize("cfg", [(2, 64, 64, 64, 128), (1, 32, 128, 128, 64), (2, 32, 32, 32, 256), (1, 64, 256, 256, 32)], ids=str)
def test_halo_forward_on_the_fp16_split(cfg, device):
    """conv_halo16_f16x3_kernel: the forward passes that run on the fp16 3-term split (the flow network, flows_generator.py:45-54,
    ops.bf16x3_min_rows(fwd_h16=True): fp32 gathered tensor, weights packed x 32 as fp16 hi / lo planes) on the halo-tile pipeline --
    against the oracle at the fp16 split's accuracy (~5e-6 of rms: bound 3e-5 of max) and the wave-specialised fp16-split kernel."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, Kout = cfg
    torch.manual_seed(29)
    x = torch.randn(N, C, H, W, device=device)
    w = torch.randn(Kout, C, 3, 3, device=device) * 0.03
    b = torch.randn(Kout, device=device)
    outs = {}
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32, fwd_h16=True), torch.no_grad():
        for on in (True, False):
            with _halo(on):
                outs[on] = ops.conv2d(x, w, b, 1, 1, relu=True)
                name = _lib.query('ag2v_last_conv_kernel')
                assert name.startswith(b'conv_halo16_f16x3_kernel' if on else b'conv_igemm_f16x3'), name
    ref = K.conv2d(x.cpu(), w.cpu(), b.cpu(), 1, 1, relu=True)
    sc = float(ref.abs().max())
    assert float((outs[True].cpu() - ref).abs().max()) <= 3e-5 * sc
    assert float((outs[True] - outs[False]).abs().max()) <= 1e-5 * sc


@pytest.mark.parametrize("cfg", [(3, 64, 64, 64, 3, True, True), (2, 32, 128, 128, 2, True, False), (2, 32, 32, 32, 3, False, False),
                                 (1, 32, 224, 224, 2, True, True)], ids=str)
def test_shared_input_group_as_one_launch_equals_separate_launches(cfg, device):
    """conv2d_shared_input (the mlp_shared convolutions of a SPADE block: same resized segmentation map, 128 output channels each --
    reference normalization.py:90-94, architecture.py:52-62) as ONE forward launch whose 128-row tiles write their own tensors
    (ag2v_conv_opts::out_mstride): bit-identical to the separate launches (same tiles, same products, same order), against the oracle
    at 2e-4, with split-word and fp32 inputs / outputs, in both MFMA shapes; the backward (merged gradients) is the same either way.
    Shapes the LDS-DMA kernels do not take keep the separate launches."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, nb, x_split, out_split = cfg
    torch.manual_seed(5)
    x = torch.randn(N, C, H, W, device=device)
    xs = _to_split_words(x)
    xin = xs if x_split else x
    xv = _from_split_words(xs) if x_split else x
    ws = [(torch.randn(128, C, 3, 3, device=device) * 0.05).requires_grad_(True) for _ in range(nb)]
    bs = [torch.randn(128, device=device).requires_grad_(True) for _ in range(nb)]
    cots = [torch.randn(N, 128, H, W, device=device) for _ in range(nb)]

    def run(merged, fam):
        ops._MERGE_SHARED_FWD = merged
        with _halo(fam), ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
            xi = xin.clone().requires_grad_(True)
            if x_split:
                ops.mark_split(xi)
            outs = ops.conv2d_shared_input(xi, ws, bs, stride=1, pad=1, relu=True, grad_premasked=False)
            name = _lib.query('ag2v_last_conv_kernel')
            for w_ in ws + bs:
                w_.grad = None
            sum((o * c).sum() for o, c in zip(outs, cots)).backward()
            return [o.detach().clone() for o in outs], xi.grad.clone(), [w_.grad.clone() for w_ in ws], name
    try:
        for fam in (True, '32'):
            sep, dx_s, dw_s, _ = run(False, fam)
            mer, dx_m, dw_m, name = run(True, fam)
            assert name.startswith(b'conv_halo'), name
            for a, b_ in zip(sep, mer):
                assert b_.is_contiguous() and torch.equal(a, b_)
            assert torch.equal(dx_s, dx_m) or float((dx_s - dx_m).abs().max()) <= 1e-6 * float(dx_s.abs().max())
            for a, b_ in zip(dw_s, dw_m):
                assert float((a - b_).abs().max()) <= 1e-6 * float(a.abs().max())
            for j in range(nb):
                ref = K.conv2d(xv.cpu(), ws[j].detach().cpu(), bs[j].detach().cpu(), 1, 1, relu=True)
                assert float((mer[j].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
        # split-word outputs (relu + premasked gradient: what the SPADE blocks ask for) through the merged launch
        if out_split:
            res = {}
            for merged in (False, True):
                ops._MERGE_SHARED_FWD = merged
                with _halo(True), ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
                    xi = ops.mark_split(xin.clone()) if x_split else xin.clone()
                    res[merged] = ops.conv2d_shared_input(xi, [w_.detach() for w_ in ws], [b_.detach() for b_ in bs], stride=1, pad=1, relu=True,
                                                          grad_premasked=True, out_split=[True] * nb)
            for a, b_ in zip(res[False], res[True]):
                assert ops.is_split(a) and ops.is_split(b_) and torch.equal(a.view(torch.int32), b_.view(torch.int32))
        # a map the LDS-DMA kernels do not take (16 x 16 at the default tile threshold): the group keeps its separate launches
        ops._MERGE_SHARED_FWD = True
        small = torch.randn(2, C, 16, 16, device=device)
        with ops.conv_precision('bf16x3'), torch.no_grad():
            assert not ops._fwd_mtile_ok(None, small, 128 * nb, 3, 3, 1, 1, False, False)
            outs = ops.conv2d_shared_input(small, [w_.detach() for w_ in ws], [b_.detach() for b_ in bs], stride=1, pad=1, relu=True)
            for j in range(nb):
                ref = K.conv2d(small.cpu(), ws[j].detach().cpu(), bs[j].detach().cpu(), 1, 1, relu=True)
                assert float((outs[j].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    finally:
        ops._MERGE_SHARED_FWD = True


def _is_wide(name):
    """the CB = 4 instantiations carry one template argument more: <WM, WN, EM, TW, TH, PS, 4> / f16x3: <WM, WN, EM, TW, TH, 4>"""
    return name.count(b',') == (5 if name.startswith(b'conv_halo16_f16x3') else 6)


@pytest.mark.parametrize("cfg", [(3, 64, 64, 64, 64), (2, 32, 128, 128, 48), (2, 96, 256, 256, 64), (1, 32, 224, 128, 32), (2, 64, 64, 192, 64)], ids=str)
def test_wide_tile_of_the_64_row_layers(cfg, device):
    """Round 6: layers with <= 64 output (forward) / input (data gradient) channels -- the decoder's up_3 block, the flow network's
    full-resolution layers: reference architecture.py:29-32, flows_generator.py:30-54 -- take 512 pixels per workgroup, 64 per wave
    (conv_halo.hip HaloCfg CB = 4: 128 x 4 or 64 x 8 patches, seven DMA rounds per stage) where the map allows it.  Against the oracle
    (2e-4) and against the 256-pixel tile they ran on before (AG2V_NO_HALO_WIDE: same products, same order per output -- 1e-6): forward
    with every epilogue on split words and on fp32 words, the fp16-split forward, the masked data gradient; maps the wide tile does not
    divide keep the narrow one."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    N, C, H, W, Kout = cfg
    torch.manual_seed(9)
    x = torch.randn(N, C, H, W, device=device)
    xs = _to_split_words(x)
    xv = _from_split_words(xs)
    w = torch.randn(Kout, C, 3, 3, device=device) * 0.05
    b = torch.randn(Kout, device=device)
    res = torch.randn(N, Kout, H, W, device=device)
    gs = torch.linspace(0.5, 1.5, N, device=device)
    wide_ok = (W % 64 == 0) and ((W % 128 == 0 and H % 4 == 0 and H > 4) or (H % 8 == 0 and H > 8))
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
        for kw in (dict(bias=b, relu=True), dict(bias=None, residual=res), dict(bias=b, gscale=gs), dict(bias=b, relu=True, split_out=True)):
            outs = {}
            for on in (True, 'narrow'):
                with _halo(on):
                    xin = ops.mark_split(xs.clone())
                    if kw.get('split_out'):
                        outs[on] = ops.conv2d_shared_input(xin, [w], [b], stride=1, pad=1, relu=True, grad_premasked=True, out_split=[True])[0]
                    else:
                        outs[on] = ops.conv2d(xin, w, kw.get('bias'), 1, 1, residual=kw.get('residual'), relu=kw.get('relu', False), gscale=kw.get('gscale'))
                    name = _lib.query('ag2v_last_conv_kernel')
                    assert name.startswith(b'conv_halo16_bf16x3_kernel<1, 8, 2'), name
                    assert _is_wide(name) == (on is True and wide_ok), (on, name)
            a, n_ = outs[True], outs['narrow']
            if kw.get('split_out'):
                assert torch.equal(a.view(torch.int32), n_.view(torch.int32))
                continue
            ref = K.conv2d(xv.cpu(), w.cpu(), None if kw.get('bias') is None else b.cpu(), 1, 1,
                           residual=None if kw.get('residual') is None else res.cpu(), relu=kw.get('relu', False),
                           gscale=None if kw.get('gscale') is None else gs.cpu())
            scale = float(ref.abs().max())
            assert float((a.cpu() - ref).abs().max()) <= 2e-4 * scale, sorted(kw)
            assert float((a - n_).abs().max()) <= 1e-6 * scale, sorted(kw)
        # fp32 gathered words, and the fp16 3-term split of the flow network's forward
        for h16 in (False, True):
            outs = {}
            for on in (True, 'narrow'):
                with _halo(on):
                    if h16:
                        with ops.bf16x3_min_rows(32, fwd_h16=True):
                            outs[on] = ops.conv2d(x, w, b, 1, 1)
                    else:
                        outs[on] = ops.conv2d(x, w, b, 1, 1)
                    name = _lib.query('ag2v_last_conv_kernel')
                    assert _is_wide(name) == (on is True and wide_ok), (on, h16, name)
                    assert name.startswith(b'conv_halo16_f16x3_kernel' if h16 else b'conv_halo16_bf16x3_kernel'), name
            ref = K.conv2d(x.cpu(), w.cpu(), b.cpu(), 1, 1)
            assert float((outs[True].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
            assert float((outs[True] - outs['narrow']).abs().max()) <= 1e-6 * float(ref.abs().max())
    # data gradient with the ReLU mask (rows = input channels <= 64): autograd through conv2d(relu_input=True)
    if C <= 64:
        wd = (torch.randn(128, C, 3, 3, device=device) * 0.05)
        dy = _to_split_words(torch.randn(N, 128, H, W, device=device))
        grads = {}
        for on in (True, 'narrow'):
            with _halo(on), ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
                xi = torch.relu(x).clone().requires_grad_(True)
                y = ops.conv2d(xi, wd, None, 1, 1, relu_input=True)
                y.backward(_from_split_words(dy))
                grads[on] = xi.grad.clone()
        xr = torch.relu(x).double().cpu().requires_grad_(True)
        torch.nn.functional.conv2d(xr, wd.double().cpu(), None, 1, 1).backward(_from_split_words(dy).double().cpu())
        gref = xr.grad * (xr.detach() > 0)
        assert float((grads[True].double().cpu() - gref).abs().max()) <= 2e-4 * float(gref.abs().max())
        assert float((grads[True] - grads['narrow']).abs().max()) <= 1e-6 * float(gref.abs().max())


def test_more_than_2048_output_channels_stay_off_the_halo_kernels(device):
    """The halo kernels keep the launch's bias vector in LDS (HaloCfg::BIAS_ROWS = 2048 + 128 rows): a wider layer runs on the
    wave-specialised kernels -- same result against the oracle."""
    from ag2video_amd import _lib, ops
    from oracle import kernels as K
    torch.manual_seed(5)
    x = torch.randn(1, 32, 32, 32, device=device)
    w = torch.randn(2176, 32, 3, 3, device=device) * 0.05
    b = torch.randn(2176, device=device)
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad(), _halo(True):
        y = ops.conv2d(x, w, b, 1, 1, relu=True)
        name = _lib.query('ag2v_last_conv_kernel')
        assert not name.startswith(b'conv_halo'), name
        y2 = ops.conv2d(x, w[:2048].contiguous(), b[:2048].contiguous(), 1, 1, relu=True)
        assert _lib.query('ag2v_last_conv_kernel').startswith(b'conv_halo16'), _lib.query('ag2v_last_conv_kernel')
    ref = K.conv2d(x.cpu(), w.cpu(), b.cpu(), 1, 1, relu=True)
    assert float((y.cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    assert float((y2.cpu() - ref[:, :2048]).abs().max()) <= 2e-4 * float(ref.abs().max())
