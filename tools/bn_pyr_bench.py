"""Backward of the batch norm + segmentation-map pyramid node at c3 size (120 x 512 x 256 x 256), with and without the level gradients:
how much of the norm_bwd_{reduce,apply}<0, true, true> time is the pyramid gather.  python tools/bn_pyr_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ag2video_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
N, C, H, W, G = 120, 512, 256, 256, 15
x = torch.randn(N, C, H, W, device=dev)
w = torch.ones(C, device=dev, requires_grad=True)
b = torch.zeros(C, device=dev, requires_grad=True)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
sizes_all = ((256, 256), (128, 128), (64, 64), (32, 32), (16, 16), (8, 8))
for sizes in (sizes_all, ((256, 256),), sizes_all[:2]):
    xr = x.clone().requires_grad_(True)
    outs = ops.batchnorm_act_pyramid(xr, w, b, rm, rv, sizes, True, 0.1, 1e-5, 0.2, G)
    grads = [torch.randn_like(o) for o in outs]
    ops.enable_kernel_timing(True)
    for rep in range(3):
        torch.autograd.backward(outs, grads, retain_graph=True)
    torch.cuda.synchronize()
    t = ops.kernel_timing_summary(by_shape=True)
    ops.enable_kernel_timing(False)
    for k, v in t.items():
        if 'norm_bwd' in k:
            print(len(sizes), 'levels:', k, '%.2f ms per call' % (v['ms'] / v['launches']), '%.2f TB/s algorithmic' % (v['flops'] / v['launches'] / (v['ms'] / v['launches'] * 1e-3) / 1e12))
    del outs, grads, xr

# forward: normalisation pass + pyramid construction, with and without the half-size level written by the normalisation pass
for env in (None, '1'):
    if env:
        os.environ['AG2V_NO_APPLY_HALF'] = env
    else:
        os.environ.pop('AG2V_NO_APPLY_HALF', None)
    with torch.no_grad():
        for rep in range(2):
            outs = ops.batchnorm_act_pyramid(x, w, b, rm, rv, sizes_all, True, 0.1, 1e-5, 0.2, G)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(5):
            outs = ops.batchnorm_act_pyramid(x, w, b, rm, rv, sizes_all, True, 0.1, 1e-5, 0.2, G)
        e1.record()
        torch.cuda.synchronize()
        print('forward (stats + apply + 5 levels), half level %s: %.2f ms' % ('by resize' if env else 'in the apply pass', e0.elapsed_time(e1) / 5))
