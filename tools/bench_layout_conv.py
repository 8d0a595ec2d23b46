"""Micro-benchmark of the low-rank layout convolution at the c3 conv_dim_in shape (120 samples, 22 objects, 512 channels, 256x256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import ops
dev = torch.device('cuda:0')
NB, J, K, H = 120, int(os.environ.get('J', '22')), int(os.environ.get('K', '512')), 256
U = torch.randn(NB, J, K, 9, device=dev)
boxes = torch.rand(NB, J, 4, device=dev) * 0.15 + 0.1
boxes[:, :, :2] = torch.rand(NB, J, 2, device=dev) * 0.7
boxes[:, J // 2 - 1] = torch.tensor([0., 0., 1., 1.], device=dev)
boxes[:, J - 1] = torch.tensor([0., 0., 1., 1.], device=dev)
keep = torch.ones(NB, J, dtype=torch.int32, device=dev)
keep[:, J // 2 - 1] = 0
keep[:, J - 1] = 0
img = torch.randn(NB, 3, H, H, device=dev)
w_img = torch.randn(K, 3, 3, 3, device=dev) * 0.1
gs = torch.linspace(0.5, 1.5, 15, device=dev)


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    print('layout_conv (no image, no residual)  %.2f ms' % timeit(lambda: ops.layout_conv(U, boxes, keep, H, H, 3, 1, 1, None, 0)))
    r = torch.randn(NB, K, H, H, device=dev)
    print('layout_conv (+ residual read)        %.2f ms' % timeit(lambda: ops.layout_conv(U, boxes, keep, H, H, 3, 1, 1, r, 0)))
    del r
    print('layout_conv_img (fused image conv)   %.2f ms' % timeit(lambda: ops.layout_conv_img(U, boxes, keep, H, H, 3, 1, img, w_img, gs)))
    print('conv2d image 3 -> %d                 %.2f ms' % (K, timeit(lambda: ops.conv2d(img, w_img, None, 1, 1, gscale=gs))))
