"""Micro-benchmark of the low-rank layout convolution at conv_dim_in's c3 shape (22 object slots, 512 channels, 120 x 256 x 256),
with all slots occupied and with 6 occupied (c3 clips: 2-3 objects per frame, two layouts)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ag2video_amd import ops
torch.manual_seed(0)
dev = 'cuda'
NB, J, K, H, W, R = 120, 22, 512, 256, 256, 3
U = torch.randn(NB, J, K, R * R, device=dev) * 0.1
dense = torch.rand(NB, J, 4, device=dev) * 0.5
dense[:, :, 2:] = dense[:, :, 2:] * 0.6 + 0.1
mask = torch.zeros(J, device=dev); mask[[0, 1, 2, 11, 12, 13]] = 1
sparse = dense * mask.view(1, J, 1)
keep = torch.ones(NB, J, dtype=torch.int32, device=dev)
img = torch.randn(NB, 3, H, W, device=dev); wimg = torch.randn(K, 3, R, R, device=dev) * 0.1
gs = torch.ones(15, device=dev)
for rep in range(2):
    for bname, boxes in (('dense', dense), ('sparse', sparse)):
        for name, fn in (('layout_conv_img', lambda: ops.layout_conv_img(U, boxes, keep, H, W, R, 1, img, wimg, gs)),
                         ('layout_conv', lambda: ops.layout_conv(U, boxes, keep, H, W, R, 1, 1, None, 0))):
            with torch.no_grad():
                y = fn(); torch.cuda.synchronize()
                t = time.time()
                for _ in range(5): y = fn()
                torch.cuda.synchronize()
                print(rep, bname, name, '%.2f ms' % ((time.time() - t) / 5 * 1e3), float(y.abs().mean()))
            del y
