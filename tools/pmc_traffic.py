"""Known-traffic launches for calibrating FETCH_SIZE / WRITE_SIZE on this access pattern + the SPADE conv / norm kernels.
Run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace ...  and again with --pmc WRITE_SIZE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import ops
dev = torch.device('cuda:0')
N, C, H = 8, 64, 256
x = torch.randn(N, C, H, H, device=dev); g = torch.randn_like(x); b = torch.randn_like(x)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
for _ in range(3):
    y = ops.spade_norm(x, g, b, rm, rv, True, 0.1, 1e-5, 0.2)          # norm_apply<1,true>: reads 3 tensors, writes 1 (16 B/lane)
    z = ops.leaky_relu(x, 0.2)                                          # act_fwd: read 1, write 1 (16 B/lane)
xs = torch.randn(8, 512, 256, 256, device=dev); w = torch.randn(128, 512, 3, 3, device=dev) * 0.01
for _ in range(3):
    o = ops.conv2d(xs, w, None, 1, 1)                                    # SPADE mlp_shared shape: algorithmic 1.07 GB in + 0.27 GB out
xr = x.clone().requires_grad_(True)
for _ in range(3):
    ops.leaky_relu(xr, 0.2).backward(g)                                  # act_bwd: reads 2 tensors, writes 1, 4 B/lane
# bf16x3 wave-specialised kernels, forward + data gradient + weight gradient:
#   SPADE mlp_shared 512 -> 128 (128x256 tile; x 1.07 GB, y / dy 0.27 GB) and a 256 -> 256 layer at 128x128 (256x128 tile; 134 MB each)
xs.requires_grad_(True); w.requires_grad_(True)
for _ in range(2):
    o2 = ops.conv2d(xs, w, None, 1, 1)
    o2.backward(torch.ones_like(o2))
xt = torch.randn(8, 256, 128, 128, device=dev, requires_grad=True); wt = (torch.randn(256, 256, 3, 3, device=dev) * 0.01).requires_grad_(True)
for _ in range(2):
    o3 = ops.conv2d(xt, wt, None, 1, 1)
    o3.backward(torch.ones_like(o3))
torch.cuda.synchronize()
print('tensor bytes', x.numel() * 4, 'conv in', xs.numel() * 4, 'conv out', o.numel() * 4)
