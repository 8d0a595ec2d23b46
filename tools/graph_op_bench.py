"""Per-operator latency inside the graph-program interpreter: programs of 100 identical operators (c3-sized operands)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import graph_program as gp
dev = torch.device('cuda:0')
B, O, T, Hd, Dp = 8, 11, 17, 512, 128
BO, BT = B * O, B * T


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


i0 = (torch.randint(0, O, (BT,), device=dev) + torch.arange(B, device=dev).repeat_interleave(T) * O).to(torch.int32)
i1 = (torch.randint(0, O, (BT,), device=dev) + torch.arange(B, device=dev).repeat_interleave(T) * O).to(torch.int32)
valid = (torch.rand(BT, device=dev) > 0.3).to(torch.uint8)
N = 100
for name in ('empty', 'gemm_88x512x512', 'gemm_136x1152x512', 'gemm_136x512x1536', 'gemm_88x128x512', 'cat_gather', 'pool', 'pool_bwd', 'gather_bwd', 'add3'):
    P = gp.Program(dev)
    if name == 'empty':
        x = P.new(8, 4)
        for _ in range(N):
            P.add3(P.new(8, 4), x)
    elif name.startswith('gemm'):
        M, Nn, K = [int(v) for v in name.split('_')[1].split('x')]
        a = P.ext(torch.randn(M, K, device=dev)); w = P.ext(torch.randn(Nn, K, device=dev))
        for _ in range(N):
            P.gemm(a, w, P.new(M, Nn), relu=True)
    elif name == 'cat_gather':
        ov = P.ext(torch.randn(BO, 512, device=dev)); pv = P.ext(torch.randn(BT, 128, device=dev))
        for _ in range(N):
            P.cat(P.new(BT, 1152), [(ov, i0), (pv, None), (ov, i1)])
    elif name == 'pool':
        nt = P.ext(torch.randn(BT, 1152, device=dev))
        for _ in range(N):
            P.pool(nt, P.new(BO, Hd), P.new(BO, 1), i0, i1, valid, O, T, Hd, Dp)
    elif name == 'pool_bwd':
        dp = P.ext(torch.randn(BO, Hd, device=dev)); cnt = P.ext(torch.ones(BO, 1, device=dev))
        for _ in range(N):
            P.pool_bwd(dp, None, P.new(BT, 1152), cnt, i0, i1, valid, O, T, Hd, Dp)
    elif name == 'gather_bwd':
        dt = P.ext(torch.randn(BT, 1152, device=dev))
        for _ in range(N):
            P.gather_bwd(dt, P.new(BO, 512), i0, i1, O, T, 512, 128)
    elif name == 'add3':
        x = P.ext(torch.randn(BO, 512, device=dev))
        for _ in range(N):
            P.add3(P.new(BO, 512), x, x, x)
    P.finalize()
    print('%-20s %.2f us per operator' % (name, timeit(P.run) * 1e3 / N))
