"""Diagnostic: every operator of the frame-batched pass at c3 sizes -- the full [120, ...] batch against the same operator applied to
each frame group of 8 samples.  Prints the worst relative difference per group (forward only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import ops
dev = torch.device('cuda:0')
T, B = 15, 8
N = T * B


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def report(name, full, parts):
    ref = torch.cat(parts, dim=0)
    errs = [((full[g * B:(g + 1) * B] - ref[g * B:(g + 1) * B]).abs().max() / ref.abs().max()).item() for g in range(T)]
    bad = [(g, round(e, 5)) for g, e in enumerate(errs) if e > 1e-4]
    print('%-44s max %.2e  %s' % (name, max(errs), ('BAD groups ' + str(bad)) if bad else 'ok'), flush=True)


def conv_case(name, C, H, K, R=3, stride=1, pad=1, exact=False, gs=True, res=False):
    x = rnd(N, C, H, H, seed=1)
    w = rnd(K, C, R, R, seed=2, scale=(C * R * R) ** -0.5)
    s = (torch.linspace(0.5, 1.5, T, device=dev)) if gs else None
    OH = (H + 2 * pad - R) // stride + 1
    r = rnd(N, K, OH, OH, seed=3) if res else None
    with ops.bf16x3_min_rows(ops.EXACT if exact else 48):
        full = ops.conv2d(x, w, None, stride, pad, residual=r, gscale=s)
        parts = [ops.conv2d(x[g * B:(g + 1) * B], w, None, stride, pad, residual=None if r is None else r[g * B:(g + 1) * B],
                            gscale=None if s is None else s[g:g + 1]) for g in range(T)]
    report(name, full, parts)
    del x, w, full, parts
    torch.cuda.empty_cache()


def bn_case(name, C, H, mode):
    x = rnd(N, C, H, H, seed=4) * 2 + 0.5
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    if mode == 'bn':
        w, b = torch.linspace(0.5, 1.5, C, device=dev), torch.linspace(-1, 1, C, device=dev)
        full = ops.batchnorm_act(x, w, b, rm, rv, True, 0.1, 1e-5, 0.2, groups=T)
        parts = [ops.batchnorm_act(x[g * B:(g + 1) * B], w, b, rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2) for g in range(T)]
    else:
        gb = rnd(N, 2 * C, H, H, seed=5, scale=0.3)
        full = ops.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, 0.2, groups=T)
        parts = [ops.spade_norm_fused(x[g * B:(g + 1) * B], gb[g * B:(g + 1) * B], rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2) for g in range(T)]
    report(name, full, parts)
    del x, full, parts
    torch.cuda.empty_cache()


which = sys.argv[1:] or ['flow', 'dec']
if 'flow' in which:
    conv_case('flow.down0 img 3->32 exact', 3, 256, 32, exact=True)
    bn_case('flow.bn 32@256', 32, 256, 'bn')
    conv_case('flow.down1 32->64 s2 exact', 32, 256, 64, stride=2, exact=True)
    conv_case('flow.down2 64->128 s2 exact', 64, 128, 128, stride=2, exact=True)
    conv_case('flow.down3 128->256 s2 exact', 128, 64, 256, stride=2, exact=True)
    conv_case('flow.res 256->256 @32 exact +res', 256, 32, 256, exact=True, res=True)
    bn_case('flow.bn 256@32', 256, 32, 'bn')
    conv_case('flow.up 256->128 @64 exact', 256, 64, 128, exact=True)
    conv_case('flow.up 128->64 @128 exact', 128, 128, 64, exact=True)
    conv_case('flow.up 64->32 @256 exact', 64, 256, 32, exact=True)
    conv_case('flow.conv_flow 32->2 @256 (no gscale)', 32, 256, 2, exact=True, gs=False)
    x = rnd(N, 64, 128, 128, seed=6)
    report('upsample2x 64@128', ops.upsample2x(x), [ops.upsample2x(x[g * B:(g + 1) * B]) for g in range(T)])
    img, fl = rnd(N, 3, 256, 256, seed=7), rnd(N, 2, 256, 256, seed=8) * 5
    report('flow_warp', ops.flow_warp(img, fl), [ops.flow_warp(img[g * B:(g + 1) * B], fl[g * B:(g + 1) * B]) for g in range(T)])
    U = rnd(N, 22, 32, 9, seed=9)
    boxes = torch.rand(N, 22, 4, device=dev) * 0.4 + 0.05
    keep = torch.ones(N, 22, dtype=torch.int32, device=dev)
    r = rnd(N, 32, 256, 256, seed=10)
    report('layout_conv K=32 +res', ops.layout_conv(U, boxes, keep, 256, 256, 3, 1, 1, r, 0),
           [ops.layout_conv(U[g * B:(g + 1) * B], boxes[g * B:(g + 1) * B], keep[g * B:(g + 1) * B], 256, 256, 3, 1, 1, r[g * B:(g + 1) * B], 0) for g in range(T)])
if 'dec' in which:
    conv_case('dec.conv_dim_in img 3->512', 3, 256, 512)
    U = rnd(N, 22, 512, 9, seed=9)
    boxes = torch.rand(N, 22, 4, device=dev) * 0.4 + 0.05
    keep = torch.ones(N, 22, dtype=torch.int32, device=dev)
    r = rnd(N, 512, 256, 256, seed=10)
    report('layout_conv K=512 +res', ops.layout_conv(U, boxes, keep, 256, 256, 3, 1, 1, r, 0),
           [ops.layout_conv(U[g * B:(g + 1) * B], boxes[g * B:(g + 1) * B], keep[g * B:(g + 1) * B], 256, 256, 3, 1, 1, r[g * B:(g + 1) * B], 0) for g in range(T)])
    del U, r
    torch.cuda.empty_cache()
    bn_case('dec.bn 512@256', 512, 256, 'bn')
    x = rnd(N, 512, 256, 256, seed=11)
    sizes = [(8 << i, 8 << i) for i in range(6)]
    full = ops.nearest_pyramid(x, sizes)
    for i, (h, w) in enumerate(sizes):
        report('pyramid level %d' % h, full[i], [ops.nearest_pyramid(x[g * B:(g + 1) * B], sizes)[i] for g in range(T)])
    del x, full
    torch.cuda.empty_cache()
    conv_case('dec.mlp_shared 512->128 @256 (no gscale)', 512, 256, 128, gs=False)
    conv_case('dec.gamma_beta 128->128 @256 (no gscale)', 128, 256, 128, gs=False)
    bn_case('dec.spade 64@256', 64, 256, 'spade')
    bn_case('dec.spade 128@256', 128, 256, 'spade')
    conv_case('dec.up3.conv_0 128->64 @256 gscale', 128, 256, 64)
    conv_case('dec.up3.conv_s 128->64 1x1 gscale', 128, 256, 64, R=1, pad=0)
    conv_case('dec.up3.conv_1 64->64 +res', 64, 256, 64, res=True)
    conv_case('dec.head 1024->1024 @8', 1024, 8, 1024)
    conv_case('dec.up0 1024->512 @32', 1024, 32, 512)
    conv_case('dec.fc 512->1024 @8 (no gscale)', 512, 8, 1024, gs=False)
    conv_case('dec.conv_img 64->3 (no gscale)', 64, 256, 3, gs=False)
