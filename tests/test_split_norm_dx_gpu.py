"""Split-word input gradients out of the SPADE backwards (ops.spade_norm_fused(x_single_reader=True), ops.spade_pair(xs_single_reader=
True), ops.conv2d(residual_single_reader=True)).

In a decoder block (reference architecture.py:40-62) conv_0's output is read by norm_1 only, a learned-shortcut block's output by the
next block's norm_0 / norm_s pair only, and conv_s's output by conv_1's residual add only.  The backward kernels of those norms then
write d(x) as split words (bf16 hi | bf16 lo), which conv_0 / conv_1 / conv_s read in their data and weight gradients without the
fp32 -> (hi, lo) split in their producers.  The words are exactly what the kernels' own split of the fp32 value produces, so the products and
sums are the same: in deterministic mode the weight / input gradients equal those of the fp32 hand-over (AG2V_NO_SPLIT_NORM_DX) to the bit
(measured 0.0; bounded at 1e-6) -- only the bias sums in front of a parameter-free batch norm, whose true value is zero, see the rounded
words.  And the split form must really be what the convolutions received."""
import copy
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _blocks(device, chans, seg_nc):
    from ag2video_amd.models.spade_models.networks.architecture import SPADEResnetBlock
    opt = types.SimpleNamespace(norm_G='spectralspadesyncbatch3x3', semantic_nc=seg_nc)
    torch.manual_seed(3)
    blks = [SPADEResnetBlock(a, b, opt).to(device).train() for a, b in zip(chans[:-1], chans[1:])]
    for blk in blks:
        for p in blk.parameters():
            if p.dim() > 1:
                p.data.mul_(1.5)
    return blks


@pytest.mark.parametrize("cfg", [(4, (128, 64, 32), 16, 2), (2, (256, 128, 64), 8, 1)], ids=str)
def test_split_input_gradients_of_the_spade_backwards(cfg, device, monkeypatch):
    from ag2video_amd import ops
    N, chans, H0, groups = cfg
    seg_nc = 64
    blks0 = _blocks(device, chans, seg_nc)
    torch.manual_seed(5)
    Hs = H0 * 2 ** (len(chans) - 2)
    seg = torch.randn(N, seg_nc, Hs, Hs, device=device)
    z = torch.randn(N, 32, H0, H0, device=device)
    w_fc = (torch.randn(chans[0], 32, 3, 3, device=device) * 0.08)
    head = torch.randn(N, chans[-1], Hs, Hs, device=device)
    real_check = ops._check_taken_offer
    seen = []

    def spy(ctx, nr, dy, what):      # every convolution's backward passes its incoming gradient through this check
        if what == 'conv2d':
            seen.append(dy is not None and ops.is_split(dy))
        return real_check(ctx, nr, dy, what)

    def run(on):
        monkeypatch.setattr(ops, '_SPLIT_NORM_DX', on)
        blks = copy.deepcopy(blks0)
        wf = w_fc.clone().requires_grad_(True)
        del seen[:]
        monkeypatch.setattr(ops, '_check_taken_offer', spy)
        try:
            with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), ops.deterministic(True), ops.frame_groups(groups):
                from ag2video_amd.models.spade_models.networks.normalization import SegmapPyramid
                pyr = SegmapPyramid(seg)
                x = ops.conv2d(z, wf, None, 1, 1)
                # the first block reads x at its own size (up = 1: x has a second reader in no branch), the others before the 2x upsampling
                x = blks[0](x, pyr, up=1, x_single_reader=True)
                for blk in blks[1:]:
                    x = blk(x, pyr, up=2, x_single_reader=True)
                loss = (x * head).sum()
                loss.backward()
        finally:
            monkeypatch.setattr(ops, '_check_taken_offer', real_check)
        grads = [wf.grad] + [p.grad for blk in blks for p in blk.parameters()]
        names = ['w_fc'] + ['blk%d.%s' % (i, n) for i, blk in enumerate(blks) for n, _ in blk.named_parameters()]
        return float(loss), grads, names, sum(seen)

    l1, g1, names, n_split1 = run(True)
    l0, g0, _, n_split0 = run(False)
    nb = len(chans) - 1
    # per block: conv_0 (from norm_1), and for every block but the last conv_1 + conv_s (from the next block's pair); + fc (from the first pair)
    assert n_split1 - n_split0 == nb + 2 * (nb - 1) + 1, (n_split1, n_split0)
    assert l1 == l0                                    # the forward pass is the same launches
    worst = 0.0
    for a, b, n in zip(g1, g0, names):
        assert a is not None and b is not None, n
        den = float(b.double().norm())
        err = float((a.double() - b.double()).norm()) / max(den, 1e-30)
        # biases in front of a parameter-free batch norm have an analytically ZERO gradient (what both paths hold is rounding noise)
        if n.endswith('conv_0.bias') or n.endswith('conv_1.bias'):
            continue
        worst = max(worst, err)
        assert err <= 1e-6, (n, err)
    print("split norm dx: %d more convolutions with a split-word dy, worst gradient deviation %.2e" % (n_split1 - n_split0, worst))


def test_a_second_reader_of_a_promised_tensor_is_refused(device):
    """x_single_reader is the caller's statement; a tensor whose gradient was promised as split words and that IS read twice must fail
    loudly (forward: second taker; backward: a summed gradient), never sum split words as floats."""
    from ag2video_amd import ops
    torch.manual_seed(1)
    N, C, H = 2, 64, 16
    x = torch.randn(N, 32, H, H, device=device)
    w = (torch.randn(C, 32, 3, 3, device=device) * 0.1).requires_grad_(True)
    gb = torch.randn(N, 2 * C, H, H, device=device)
    rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
        y = ops.conv2d(x, w, None, 1, 1)
        a = ops.spade_norm_fused(y, gb, rm, rv, True, 0.1, 1e-5, 0.2, x_single_reader=True)
        with pytest.raises(RuntimeError, match="second consumer"):
            ops.spade_norm_fused(y, gb, rm, rv, True, 0.1, 1e-5, 0.2, x_single_reader=True)
        with pytest.raises(RuntimeError, match="second consumer|split"):
            (a.sum() + (y * y).sum()).backward()
