"""Forward time of single convolutions through ops.conv2d.  usage: conv_bench.py <precision> N C H W K [R]   (precision: fp32 | bf16x3)"""
import sys, time
import torch
sys.path.insert(0, '.')
from ag2video_amd import ops, _lib
prec = sys.argv[1]
N, C, H, W, K = [int(v) for v in sys.argv[2:7]]
R = int(sys.argv[7]) if len(sys.argv) > 7 else 3
torch.manual_seed(0)
x = torch.randn(N, C, H, W, device='cuda'); w = torch.randn(K, C, R, R, device='cuda') * 0.02
with ops.conv_precision(prec), torch.no_grad():
    y = ops.conv2d(x, w, None, 1, R // 2)
    ref = torch.nn.functional.conv2d(x[:2].double(), w.double(), None, 1, R // 2).float()
    err = float((y[:2] - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        y = ops.conv2d(x, w, None, 1, R // 2)
    torch.cuda.synchronize()
    ms = (time.time() - t) / 20 * 1e3
print('%s %s: %.3f ms  %.1f TF  err %.1e  [%s]' % (prec, (N, C, H, W, K, R), ms, 2.0 * N * K * H * W * C * R * R / ms / 1e9, err,
      _lib.query('ag2v_last_conv_kernel')))
