"""The LDS-DMA weight-gradient kernel (csrc/conv_wgrad_halo.hip: 3x3 stride-1 weight gradients whose operands are split words -- the
SPADE decoder's mlp_shared / gamma | beta / conv_0 / conv_1 layers, reference normalization.py:90-94, architecture.py:29-32) against
(a) fp64 autograd of nn.Conv2d on the split words' exact values (2e-5 of max: bf16x3 products, another summation order) and (b) the
wave-specialised 16-byte-load kernels it replaces (AG2V_NO_WGRAD_HALO=1: the same products, 1e-5): every strip position (left / interior /
right image edge, one-strip-wide maps), rows beyond M in the last row tile, several images per workgroup and strips split over workgroups,
sample groups (per-frame-group gradients), accumulation onto an existing gradient, and the deterministic mode (bit-identical runs)."""
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


class _wgrad_halo(object):
    def __init__(self, on):
        self.on = on

    def _set(self, off):
        from ag2video_amd import ops
        if off:
            os.environ['AG2V_NO_WGRAD_HALO'] = '1'
        else:
            os.environ.pop('AG2V_NO_WGRAD_HALO', None)
        torch.cuda.synchronize()
        ops.reload_kernel_switches()

    def __enter__(self):
        self.old = os.environ.get('AG2V_NO_WGRAD_HALO')
        self._set(not self.on)

    def __exit__(self, *exc):
        self._set(self.old is not None)
        return False


def _wgrad(dys, xs, K, groups=1, acc_into=None, x_split=True, dy_split=True, workspace=True):
    from ag2video_amd import _lib, ops
    C = xs.shape[1]
    shape = ((groups,) if groups > 1 else ()) + (K, C, 3, 3)
    dw = acc_into.clone() if acc_into is not None else torch.full(shape, float('nan'), device=xs.device)
    with ops.conv_precision('bf16x3'), torch.no_grad():
        ops._conv_wgrad_launch(32, dys, xs, dw, K, 3, 3, 1, 1, 1 if acc_into is not None else 0, groups=groups, in_split=x_split, dy_split=dy_split,
                               workspace=workspace)
    return dw, _lib.query('ag2v_last_conv_kernel')


def _ref64(dys, xs, K, groups=1):
    x = _from_split_words(xs).double()
    dy = _from_split_words(dys).double()
    N = x.shape[0]
    outs = []
    for g in range(groups):
        sl = slice(g * (N // groups), (g + 1) * (N // groups))
        w = torch.zeros(K, x.shape[1], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
        y = torch.nn.functional.conv2d(x[sl], w, None, 1, 1)
        y.backward(dy[sl])
        outs.append(w.grad)
    return torch.stack(outs) if groups > 1 else outs[0]


# (N, C, H, W, K, groups)
CASES = [(2, 64, 16, 32, 128, 1),        # one strip per image row: left AND right edge in one strip
         (3, 64, 32, 64, 128, 1),        # two strips: a left and a right one
         (2, 128, 16, 128, 96, 1),       # four strips (interior ones), two channel tiles, 96 rows on a 128-row tile
         (5, 64, 24, 96, 200, 1),        # W = 3 strips, H not a power of two, 200 rows = two row tiles (the second with 72 rows)
         (6, 64, 16, 64, 128, 3),        # three sample groups
         (16, 128, 64, 64, 384, 1),      # enough strips for several workgroups per tile
         (2, 64, 224, 224, 128, 1)]      # c5's 224 = 7 strips


@pytest.mark.parametrize("cfg", CASES, ids=str)
def test_wgrad_halo_matches_fp64_and_previous_kernels(cfg, device):
    N, C, H, W, K, groups = cfg
    g = torch.Generator().manual_seed(7)
    xs = _to_split_words(torch.randn(N, C, H, W, generator=g).to(device))
    dys = _to_split_words(torch.randn(N, K, H, W, generator=g).to(device))
    ref = _ref64(dys, xs, K, groups)
    with _wgrad_halo(True):
        new, name = _wgrad(dys, xs, K, groups)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>', name
    with _wgrad_halo(False):
        old, name_o = _wgrad(dys, xs, K, groups)
        assert name_o.startswith(b'conv_wgrad_bf16x3_v'), name_o
    scale = ref.abs().max().item()
    assert torch.isfinite(new).all()
    assert (new.double() - ref).abs().max().item() <= 2e-5 * scale, ((new.double() - ref).abs().max().item() / scale)
    assert (new - old).abs().max().item() <= 1e-5 * scale, ((new - old).abs().max().item() / scale)


@pytest.mark.parametrize("x_split,dy_split", [(True, False), (False, True), (False, False)], ids=['x_split_dy_fp32', 'x_fp32_dy_split', 'both_fp32'])
@pytest.mark.parametrize("cfg", [(3, 64, 32, 64, 128, 1), (5, 64, 24, 96, 200, 1), (6, 128, 16, 64, 128, 3)], ids=str)
def test_wgrad_halo_with_one_fp32_operand(cfg, x_split, dy_split, device):
    """conv_0 / conv_1 / conv_s of the decoder blocks hand over x as split words and dy as fp32 (the gradient of a residual sum), other
    layers the reverse, the flow network's residual blocks both as fp32: the kernel's regrouping step then performs the hi / lo decomposition of the fp32 operand itself (the arithmetic
    of conv_common.h split2, as in the kernels it replaces)."""
    N, C, H, W, K, groups = cfg
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, C, H, W, generator=g).to(device)
    dy = torch.randn(N, K, H, W, generator=g).to(device)
    xs, dys = (_to_split_words(x) if x_split else x), (_to_split_words(dy) if dy_split else dy)
    # fp64 reference on the values the kernel sees: a split-word operand carries hi + lo (16 + 8 bits), an fp32 one is split inside
    xe = _from_split_words(xs) if x_split else _from_split_words(_to_split_words(x))
    dye = _from_split_words(dys) if dy_split else _from_split_words(_to_split_words(dy))
    ref = _ref64(_to_split_words(dye), _to_split_words(xe), K, groups)
    with _wgrad_halo(True):
        new, name = _wgrad(dys, xs, K, groups, x_split=x_split, dy_split=dy_split)
        assert name == ('conv_wgrad_halo_bf16x3_kernel<%s, %s>' % (str(dy_split).lower(), str(x_split).lower())).encode(), name
    with _wgrad_halo(False):
        old, name_o = _wgrad(dys, xs, K, groups, x_split=x_split, dy_split=dy_split)
        assert name_o.startswith(b'conv_wgrad_bf16x3_v'), name_o
    scale = ref.abs().max().item()
    assert torch.isfinite(new).all()
    assert (new.double() - ref).abs().max().item() <= 2e-5 * scale, ((new.double() - ref).abs().max().item() / scale)
    assert (new - old).abs().max().item() <= 1e-5 * scale, ((new - old).abs().max().item() / scale)


def test_wgrad_halo_accumulates_and_is_deterministic_on_request(device):
    from ag2video_amd import ops
    N, C, H, W, K = 8, 64, 32, 64, 128
    g = torch.Generator().manual_seed(11)
    xs = _to_split_words(torch.randn(N, C, H, W, generator=g).to(device))
    dys = _to_split_words(torch.randn(N, K, H, W, generator=g).to(device))
    base = torch.randn(K, C, 3, 3, generator=g).to(device)
    ref = _ref64(dys, xs, K)
    with _wgrad_halo(True):
        acc, name = _wgrad(dys, xs, K, acc_into=base)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>'
        assert (acc.double() - (ref + base.double())).abs().max().item() <= 2e-5 * ref.abs().max().item()
        with ops.deterministic():
            d1, name = _wgrad(dys, xs, K)
            d2, _ = _wgrad(dys, xs, K)
            d3, _ = _wgrad(dys, xs, K, acc_into=base)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>'
        assert torch.equal(d1, d2)
        assert (d1.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        assert (d3.double() - (ref + base.double())).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wgrad_halo_at_the_benchmarked_layer_size(device):
    """The merged mlp_shared weight gradient of a c3 step (512 -> 3 x 128 channels at 256 x 256; 6 images here): the partition into runs
    of ring rows that start and end inside strips, 24 tiles x many splits, slab reduction -- against the 16-byte-load kernels (1e-5 of
    max) and, on a window of the filter, fp64."""
    N, C, H, W, K = 6, 512, 256, 256, 384
    g = torch.Generator().manual_seed(3)
    xs = _to_split_words(torch.randn(N, C, H, W, generator=g).to(device))
    dys = _to_split_words(torch.randn(N, K, H, W, generator=g).to(device))
    with _wgrad_halo(True):
        new, name = _wgrad(dys, xs, K)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>', name
        new2, _ = _wgrad(dys, xs, K)
        assert torch.equal(new, new2)              # (slab reduction: bit-reproducible in the default mode too)
    with _wgrad_halo(False):
        old, name_o = _wgrad(dys, xs, K)
        assert name_o.startswith(b'conv_wgrad_bf16x3_vps'), name_o
    scale = old.abs().max().item()
    assert torch.isfinite(new).all()
    assert (new - old).abs().max().item() <= 1e-5 * scale
    ref = _ref64(dys[:, 5:9], xs[:, 100:108], 4)          # dW[5:9, 100:108]
    assert (new[5:9, 100:108].double() - ref).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("cfg", [(8, 64, 32, 64, 128, 1), (6, 64, 16, 64, 128, 3), (16, 128, 64, 64, 200, 1)], ids=str)
def test_wgrad_halo_atomic_epilogue_without_workspace(cfg, device):
    """A caller of ag2v_conv2d_wgrad_ex that hands over NO slab workspace (the C ABI allows it outside deterministic mode): the
    LDS-DMA kernel's pixel-range splits then meet in fp32 atomics on dW behind the launcher's memset -- per sample group, with and
    without accumulation onto an existing gradient (this path had no test; the product itself always passes the
    workspace).  Against fp64 and against the slab path; deterministic mode refuses the missing workspace."""
    from ag2video_amd import ops
    N, C, H, W, K, groups = cfg
    g = torch.Generator().manual_seed(13)
    xs = _to_split_words(torch.randn(N, C, H, W, generator=g).to(device))
    dys = _to_split_words(torch.randn(N, K, H, W, generator=g).to(device))
    base = torch.randn(*(((groups,) if groups > 1 else ()) + (K, C, 3, 3)), generator=g).to(device)
    ref = _ref64(dys, xs, K, groups)
    scale = ref.abs().max().item()
    with _wgrad_halo(True):
        slab, name = _wgrad(dys, xs, K, groups)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>', name
        atom, name = _wgrad(dys, xs, K, groups, workspace=False)
        assert name == b'conv_wgrad_halo_bf16x3_kernel<true, true>', name
        atom_acc, _ = _wgrad(dys, xs, K, groups, acc_into=base, workspace=False)
        assert torch.isfinite(atom).all()
        assert (atom.double() - ref).abs().max().item() <= 2e-5 * scale
        assert (atom - slab).abs().max().item() <= 1e-5 * scale
        assert (atom_acc.double() - (ref + base.double())).abs().max().item() <= 2e-5 * scale
        with ops.deterministic():
            _wgrad(dys, xs, K, groups)
            if any(v > 0 for k, v in ops._DET_WS_FLOATS.items() if k[:5] == (N, C, H, W, K) and k[-1] == 1):   # splits -> slabs needed
                with pytest.raises(RuntimeError):
                    _wgrad(dys, xs, K, groups, workspace=False)
