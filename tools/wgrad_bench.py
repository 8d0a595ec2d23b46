"""Weight-gradient kernels on split-word operands at c3 layer shapes (frame-batched chunk sizes): time / TF of the launch the library
chooses.  usage: wgrad_bench.py [reps]   (GPU box only)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ag2video_amd import ops, _lib

SHAPES = [(21, 512, 256, 256, 384), (60, 128, 256, 256, 256), (120, 128, 128, 128, 512), (120, 512, 128, 128, 384), (60, 128, 256, 256, 128),
          (60, 128, 256, 256, 64), (60, 64, 256, 256, 64)]


def to_split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return ((hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)).view(torch.float32)


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda')
for (N, C, H, W, K) in SHAPES:
    xs = to_split(torch.randn(N, C, H, W, device=dev))
    dys = to_split(torch.randn(N, K, H, W, device=dev))
    dw = torch.empty(K, C, 3, 3, device=dev)
    with ops.conv_precision('bf16x3'), torch.no_grad():
        run = lambda: ops._conv_wgrad_launch(32, dys, xs, dw, K, 3, 3, 1, 1, 0, in_split=True, dy_split=True)
        run(); run()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                run()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
    print('wgrad %-28s %7.3f ms %6.1f TF  %s' % ((N, C, H, W, K), best, 2.0 * N * K * H * W * C * 9 / best / 1e9, _lib.query('ag2v_last_conv_kernel').decode()), flush=True)
    del xs, dys
    torch.cuda.empty_cache()
