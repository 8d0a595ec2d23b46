"""Latency of the graph nets' GEMM shapes (c3) through the C ABI: one-round-trip kernel vs the K-loop kernels
(AG2V_GEMM_ONESHOT=0 in a second process).  Checks each result against torch."""
import sys, time
import torch
sys.path.insert(0, '.')
from ag2video_amd import _lib, ops
dev = 'cuda'
P = ops._ptr
shapes = [(80, 512, 512), (88, 512, 512), (80, 128, 512), (128, 512, 1152), (128, 1152, 512), (88, 512, 128), (120, 1152, 512),
          (120, 512, 384), (48, 512, 1152), (80, 512, 516), (80, 4, 512)]
tot = 0.0
for (M, N, K) in shapes:
    for mode in ('fwd', 'dx', 'dw'):
        torch.manual_seed(1)
        if mode == 'fwd':      # y = x w^T + b
            x = torch.randn(M, K, device=dev); w = torch.randn(N * K + 2, device=dev)[2:].view(N, K); bias = torch.randn(N, device=dev)   # (w 8 bytes off a 16-byte boundary)
            y = torch.empty(M, N, device=dev)
            args = (P(x), P(w), P(bias), P(y), M, N, K, K, 1, 1, K, N, 1, 0)
            ref = torch.relu(x.double() @ w.double().t() + bias.double()).float()
        elif mode == 'dx':     # dx = g w   (g [M,N], w [N,K])
            g = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, K, device=dev)
            args = (P(g), P(w), None, P(y), M, K, N, N, 1, K, 1, K, 0, 0)
            ref = (g.double() @ w.double()).float()
        else:                  # dw += g^T x
            g = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); y = torch.ones(N, K, device=dev)
            args = (P(g), P(x), None, P(y), N, K, M, 1, N, K, 1, K, 0, 1)
            ref = (g.double().t() @ x.double()).float() + 1.0
        st = ops._stream()
        _lib.call('ag2v_gemm_f32', *args, st)
        torch.cuda.synchronize()
        err = float((y - ref).abs().max() / ref.abs().max())
        if mode == 'dw':
            args = args[:-1] + (0,)
        n = 200
        t = time.time()
        for _ in range(n):
            _lib.call('ag2v_gemm_f32', *args, st)
        torch.cuda.synchronize()
        us = (time.time() - t) / n * 1e6
        tot += us
        print('%-4s M=%4d N=%4d K=%4d  %6.1f us  err %.1e' % (mode, M, N, K, us, err))
        assert err < 1e-5, err
print('sum %.1f us' % tot)
