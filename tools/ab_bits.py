"""Checksum of bf16x3 convolution results (forward, data gradient, weight gradient) on seeded inputs: run under two builds
(AG2V_LIB_PATH=...) to see whether a kernel change is bit-neutral.  Inputs include denormals and huge values."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ag2video_amd import ops

torch.manual_seed(0)
dev = 'cuda'
x = torch.randn(4, 128, 64, 64, device=dev)
x[0, 0, 0, :8] = torch.tensor([1e-40, -1e-40, 3e38, -3e38, 1e-38, 0.0, -0.0, 65504.0], device=dev)
w = torch.randn(256, 128, 3, 3, device=dev) * 0.05
x.requires_grad_(True); w.requires_grad_(True)
y = ops.conv2d(x, w, None, 1, 1)
g = torch.randn_like(y)
gx, gw = torch.autograd.grad(y, (x, w), g)
for n, t in (('y', y), ('dx', gx), ('dw', gw)):
    print(n, hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16], float(t.detach().float().abs().max()))
