"""Few-row weight gradients (<= 4 output channels, or <= 4 input channels with the operands swapped): conv_wgrad_fewrow_kernel against
the swapped / plain GEMM launches on the c3 shapes (conv_img, conv_flow, the flow network's first layer, D's last layer)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ag2video_amd import _lib, ops


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


dev = 'cuda'
for (N, C, H, W, K, R, pad, G) in ((120, 64, 256, 256, 3, 3, 1, 1), (120, 32, 256, 256, 2, 3, 1, 1), (120, 512, 34, 34, 1, 4, 2, 1), (120, 32, 256, 256, 3, 3, 1, 15)):
    x = torch.randn(N, C, H, W, device=dev)
    OH = H + 2 * pad - R + 1
    dy = torch.randn(N, K, OH, OH, device=dev)
    dw = torch.empty((G,) * (G > 1) + (K, C, R, R), device=dev)
    with torch.no_grad():
        ms = t(lambda: ops._conv_wgrad_launch(32, dy, x, dw, K, R, R, 1, pad, 0, groups=G))
        name = _lib.query('ag2v_last_conv_kernel')
        line = '%s direct %.3f ms (%s)' % ((N, C, H, W, K, R, G), ms, name.decode())
        if OH == H:
            dws = torch.empty_like(dw)
            # the swapped GEMM: x as the dense rows (C), dy gathered
            ms2 = t(lambda: ops._conv_wgrad_swapped(32, dy, x, dws, K, R, R, pad, 0, G, False, False))
            line += '   swapped %.3f ms (%s)  maxdiff %.2e' % (ms2, _lib.query('ag2v_last_conv_kernel').decode(), float((dw - dws).abs().max() / dw.abs().max()))
        print(line)
