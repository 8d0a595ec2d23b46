"""Launches of the dominant kernels on the shapes they see in a frame-batched c3 step, for HBM-traffic counters.
usage: pmc_traffic.py <case>   (one case per process so that a kernel name identifies the launches of interest)
Run under rocprofv3 --pmc FETCH_SIZE --kernel-trace ... and again with --pmc WRITE_SIZE (tools/pmc_quick.sh); writes the
algorithmic byte counts of the case to gpurun_out/pmc_expect_<case>.json for tools/pmc_summary.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import ops
dev = torch.device('cuda:0')
expect = {}
REPS = 2
CASE = sys.argv[1]


def split_words(t):
    """(bf16(v) << 16) | bf16(v - bf16(v)) per value (ops.is_split format), built with torch ops."""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)
    return ops.mark_split(w.view(torch.float32))


def conv_case(tag, N, C, H, K, what, split=False, precision=None):
    """what: 'fwd' | 'dgrad' | 'wgrad' of a 3x3 stride-1 convolution x[N,C,H,H] -> y[N,K,H,H], with the decoder's row threshold.
    split: the operands the step hands over as split words are split words here too (x in fwd / wgrad, dy in dgrad / wgrad): the
    *_ps / *_vps kernels."""
    x = torch.randn(N, C, H, H, device=dev)
    if split and what != 'dgrad':
        x = split_words(x)
    w = (torch.randn(K, C, 3, 3, device=dev) * 0.01)
    xb, yb, wb = N * C * H * H * 4, N * K * H * H * 4, K * C * 9 * 4
    import contextlib
    with ops.bf16x3_min_rows(48), (ops.conv_precision(precision) if precision else contextlib.nullcontext()):
        if what == 'fwd':
            for _ in range(REPS):
                ops.conv2d(x, w, None, 1, 1)
            alg = xb + yb + wb
        else:
            x.requires_grad_(what == 'dgrad')
            w.requires_grad_(what == 'wgrad')
            for _ in range(REPS):
                y = ops.conv2d(x, w, None, 1, 1)
                g = torch.ones_like(y)
                if split:
                    g = split_words(g)
                y.backward(g)
            alg = (yb + xb + wb) if what == 'dgrad' else (yb + xb + wb)
    torch.cuda.synchronize()
    expect[tag] = {'shape': '%s of %dx%dx%dx%d -> %d channels, 3x3%s' % (what, N, C, H, H, K, ', split-word operands' if split else ''),
                   'algorithmic_bytes': alg, 'pass': what}


def shared_fwd_case(tag, N, C, H, nb):
    """Round 6: the nb mlp_shared convolutions of a SPADE block (128 output channels each, one input) as ONE forward launch
    (ops.conv2d_shared_input, ag2v_conv_opts::out_mstride): the gathered map is staged once for all of them."""
    x = split_words(torch.randn(N, C, H, H, device=dev))
    ws = [torch.randn(128, C, 3, 3, device=dev) * 0.01 for _ in range(nb)]
    bs = [torch.zeros(128, device=dev) for _ in range(nb)]
    with ops.bf16x3_min_rows(48), torch.no_grad():
        for _ in range(REPS):
            ops.conv2d_shared_input(x, ws, bs, stride=1, pad=1, relu=True, grad_premasked=True, out_split=[True] * nb)
    torch.cuda.synchronize()
    expect[tag] = {'shape': 'fwd of %dx%dx%dx%d -> %d x 128 channels in one launch, 3x3, split-word operands' % (N, C, H, H, nb),
                   'algorithmic_bytes': N * C * H * H * 4 + nb * N * 128 * H * H * 4 + nb * 128 * C * 9 * 4, 'pass': 'fwd'}


# SPADE mlp_shared at full resolution: forward (128 x 256 tile, quad loads), merged data gradient of a block's three mlp_shared
# convolutions (256 x 128 tile) and their merged weight gradient (128-row v kernel); sample counts = the chunks ag2v_conv2d_*_ex cuts
CASES = {
    'mlp_shared_fwd': lambda: conv_case('mlp_shared_fwd', 31, 512, 256, 128, 'fwd'),
    'mlp_shared_dgrad_merged': lambda: conv_case('mlp_shared_dgrad_merged', 40, 512, 256, 384, 'dgrad'),
    'mlp_shared_wgrad_merged': lambda: conv_case('mlp_shared_wgrad_merged', 21, 512, 256, 384, 'wgrad'),
    'conv_256_wgrad_128px': lambda: conv_case('conv_256_wgrad_128px', 120, 256, 128, 256, 'wgrad'),
    # the same launches as the step issues them since the split-word hand-over
    'mlp_shared_fwd_ps': lambda: conv_case('mlp_shared_fwd_ps', 31, 512, 256, 128, 'fwd', True),
    'mlp_shared_dgrad_merged_ps': lambda: conv_case('mlp_shared_dgrad_merged_ps', 40, 512, 256, 384, 'dgrad', True),
    'mlp_shared_wgrad_merged_ps': lambda: conv_case('mlp_shared_wgrad_merged_ps', 21, 512, 256, 384, 'wgrad', True),
    'conv_256_wgrad_128px_ps': lambda: conv_case('conv_256_wgrad_128px_ps', 120, 256, 128, 256, 'wgrad', True),
    # round 5: the exact-fp32 leg's largest kernel, and the gamma | beta layer's passes at full resolution
    'mlp_shared_wgrad_merged_fp32': lambda: conv_case('mlp_shared_wgrad_merged_fp32', 21, 512, 256, 384, 'wgrad', False, 'fp32'),
    'gamma_beta_wgrad_ps': lambda: conv_case('gamma_beta_wgrad_ps', 60, 128, 256, 256, 'wgrad', True),
    'gamma_beta_dgrad_ps': lambda: conv_case('gamma_beta_dgrad_ps', 60, 128, 256, 256, 'dgrad', True),
    'gamma_beta_fwd_ps': lambda: conv_case('gamma_beta_fwd_ps', 60, 128, 256, 256, 'fwd', True),
    'mlp_shared_fwd_merged3_ps': lambda: shared_fwd_case('mlp_shared_fwd_merged3_ps', 31, 512, 256, 3),
}


def norm_case():
    # SPADE modulation (HBM-bound): forward and backward at up_3's size, frame groups of 8
    N, C, H = 120, 64, 256
    x = torch.randn(N, C, H, H, device=dev, requires_grad=True)
    gb = torch.randn(N, 2 * C, H, H, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    for _ in range(REPS):
        y = ops.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, 0.2, groups=15)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    tb = N * C * H * H * 4
    expect['norm_apply'] = {'shape': 'spade_norm_fused forward %dx%dx%dx%d, 15 groups' % (N, C, H, H), 'algorithmic_bytes': 4 * tb, 'pass': 'fwd'}
    expect['norm_bwd_reduce'] = {'shape': 'same, backward reduce pass', 'algorithmic_bytes': 4 * tb, 'pass': 'bwd'}
    expect['norm_bwd_apply'] = {'shape': 'same, backward apply pass', 'algorithmic_bytes': 7 * tb, 'pass': 'bwd'}
    expect['bn_stats_partial'] = {'shape': 'same, statistics pass', 'algorithmic_bytes': tb, 'pass': 'fwd'}


def pair_case():
    # SPADE pair at up_3's size: x [120, 128, 128, 128] read before its 2x upsampling, two gamma|beta tensors [120, 256, 256, 256]
    N, C, Hs = 120, 128, 128
    xs = torch.randn(N, C, Hs, Hs, device=dev, requires_grad=True)
    gb0 = torch.randn(N, 2 * C, 2 * Hs, 2 * Hs, device=dev, requires_grad=True)
    gb1 = torch.randn(N, 2 * C, 2 * Hs, 2 * Hs, device=dev, requires_grad=True)
    bufs = [torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)]
    for _ in range(REPS):
        y0, y1 = ops.spade_pair(xs, gb0, gb1, bufs[0], bufs[1], bufs[2], bufs[3], 2, 0.1, 1e-5, 0.2, 1.0, 15)
        (y0.sum() + y1.sum()).backward()
    torch.cuda.synchronize()
    ob, xb = N * C * 4 * Hs * Hs * 4, N * C * Hs * Hs * 4
    expect['spade_pair_apply'] = {'shape': 'spade_pair up=2, x %dx%dx%dx%d' % (N, C, Hs, Hs), 'algorithmic_bytes': xb + 6 * ob, 'pass': 'fwd'}
    expect['spade_pair_bwd_reduce'] = {'shape': 'same, backward reduce pass', 'algorithmic_bytes': xb + 5 * ob, 'pass': 'bwd'}
    expect['spade_pair_bwd_apply'] = {'shape': 'same, backward apply pass', 'algorithmic_bytes': 2 * xb + 9 * ob, 'pass': 'bwd'}


def layout_case():
    # conv_dim_in's low-rank layout convolution + image channels at c3 (22 object slots, 6 occupied): one 16 GB write
    NB, J, K, H, R = 120, 22, 512, 256, 3
    U = torch.randn(NB, J, K, R * R, device=dev) * 0.1
    boxes = torch.rand(NB, J, 4, device=dev) * 0.5
    boxes[:, :, 2:] = boxes[:, :, 2:] * 0.6 + 0.1
    m = torch.zeros(J, device=dev); m[[0, 1, 2, 11, 12, 13]] = 1
    boxes = boxes * m.view(1, J, 1)
    keep = torch.ones(NB, J, dtype=torch.int32, device=dev)
    img = torch.randn(NB, 3, H, H, device=dev); wimg = torch.randn(K, 3, R, R, device=dev) * 0.1
    gs = torch.ones(15, device=dev)
    with torch.no_grad():
        for _ in range(REPS):
            ops.layout_conv_img(U, boxes, keep, H, H, R, 1, img, wimg, gs)
    torch.cuda.synchronize()
    expect['layout_conv_fwd'] = {'shape': 'layout_conv_img %dx%dx%dx%d, %d object slots (6 occupied), 3 image channels' % (NB, K, H, H, J),
                                 'algorithmic_bytes': (NB * K * H * H + NB * 3 * H * H + NB * J * K * R * R) * 4, 'pass': 'fwd'}


def f32_case():
    # a residual-block convolution of the flow network (exact fp32 forward): 120 x 256 x 32 x 32 -> 256
    N, C, H, K = 120, 256, 32, 256
    x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    with ops.conv_precision('fp32'), torch.no_grad():
        for _ in range(REPS):
            ops.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    expect['flow_res_fp32'] = {'shape': 'exact fp32 fwd of %dx%dx%dx%d -> %d channels, 3x3' % (N, C, H, H, K),
                               'algorithmic_bytes': (N * C * H * H + N * K * H * H + K * C * 9) * 4, 'pass': 'fwd'}


def bn_pyramid_case():
    # conv_dim_in's batch norm + segmentation-map pyramid (one autograd node): the backward gathers the five coarser levels' gradients
    # while it loads the full-resolution one -- norm_bwd_reduce_kernel<0, true, true> / norm_bwd_apply_kernel<0, true, true>
    N, C, H = 120, 512, 256
    x = torch.randn(N, C, H, H, device=dev, requires_grad=True)
    w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sizes = [(H >> i, H >> i) for i in range(6)]
    for _ in range(REPS):
        outs = ops.batchnorm_act_pyramid(x, w, b, rm, rv, sizes, True, 0.1, 1e-5, 0.2, groups=15)
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
        x.grad = None
    torch.cuda.synchronize()
    tb = N * C * H * H * 4
    lv = tb / 3.0 * (1 - 4.0 ** -5)          # the five coarser levels: 1/4 + 1/16 + ... of the full tensor
    expect['bn_pyr_bwd_reduce'] = {'shape': 'batch norm + 6-level pyramid backward, reduce pass, %dx%dx%dx%d, 15 groups' % (N, C, H, H),
                                   'algorithmic_bytes': 2 * tb + lv, 'pass': 'bwd'}
    expect['bn_pyr_bwd_apply'] = {'shape': 'same, apply pass', 'algorithmic_bytes': 3 * tb + lv, 'pass': 'bwd'}


def layout_bwd_case():
    # backward of conv_dim_in's low-rank layout convolution: reads each object's support rectangle of the 16 GB output gradient
    NB, J, K, H, R = 120, 22, 512, 256, 3
    U = (torch.randn(NB, J, K, R * R, device=dev) * 0.1).requires_grad_(True)
    boxes = torch.rand(NB, J, 4, device=dev) * 0.5
    boxes[:, :, 2:] = boxes[:, :, 2:] * 0.6 + 0.1
    m = torch.zeros(J, device=dev); m[[0, 1, 2, 11, 12, 13]] = 1
    boxes = boxes * m.view(1, J, 1)
    area = float((boxes[:, :, 2] * boxes[:, :, 3]).sum() / NB)        # mean covered fraction of the image, summed over the objects
    keep = torch.ones(NB, J, dtype=torch.int32, device=dev)
    g = torch.randn(NB, K, H, H, device=dev)
    for _ in range(REPS):
        y = ops.layout_conv(U, boxes, keep, H, H, R, 1, 1)
        y.backward(g)
        U.grad = None
    torch.cuda.synchronize()
    expect['layout_conv_bwd'] = {'shape': 'layout_conv backward %dx%dx%dx%d, %d object slots (6 occupied, mean covered area %.2f of the image)' % (NB, K, H, H, J, area),
                                 'algorithmic_bytes': (area * NB * K * H * H + NB * J * K * R * R) * 4, 'pass': 'bwd'}


def h16_case():
    # the same flow-network layer on the fp16 3-term split (the product's forward arithmetic there since round 3)
    N, C, H, K = 120, 256, 32, 256
    x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    with ops.bf16x3_min_rows(32, 32, fwd_h16=True), torch.no_grad():
        for _ in range(REPS):
            ops.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    expect['flow_res_h16'] = {'shape': 'fp16-split fwd of %dx%dx%dx%d -> %d channels, 3x3' % (N, C, H, H, K),
                              'algorithmic_bytes': (N * C * H * H + N * K * H * H + K * C * 9) * 4, 'pass': 'fwd'}


CASES['flow_res_h16'] = h16_case
CASES['bn_pyramid'] = bn_pyramid_case
CASES['layout_conv_bwd'] = layout_bwd_case
CASES['layout_conv'] = layout_case
CASES['flow_res_fp32'] = f32_case
CASES['norm'] = norm_case
CASES['spade_pair'] = pair_case
CASES[CASE]()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
json.dump(expect, open(os.path.join(root, 'gpurun_out', 'pmc_expect_%s.json' % CASE), 'w'), indent=1)
print('done')
