import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ag2video_amd import ops
from test_ops_gpu import _to_split_words
dev = 'cuda'
torch.manual_seed(5)
N, C, H, W, K, R, pad = 2, 64, 32, 32, 128, 3, 1
for special in (False, True):
    x = torch.randn(N, C, H, W, device=dev)
    if special:
        x[0, 0, 0, :6] = torch.tensor([0.0, -0.0, 1e-30, -3e20, 1.0, 65504.0], device=dev)
    w = (torch.randn(K, C, R, R, device=dev) * 0.05)
    with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32):
        y0 = ops.conv2d(x.clone(), w, None, 1, pad)
        k0 = ops._lib.query('ag2v_last_conv_kernel')
        y1 = ops.conv2d(ops.mark_split(_to_split_words(x)), w, None, 1, pad)
    d = (y0 - y1).abs()
    print('special', special, 'max', float(d.max()), 'n diff', int((d > 0).sum()), 'of', d.numel())
    idx = torch.nonzero(d > 0)[:8]
    for i in idx:
        n, k, yy, xx = [int(v) for v in i]
        print('  at', (n, k, yy, xx), float(y0[n, k, yy, xx]), float(y1[n, k, yy, xx]))
C = 64
x = torch.randn(4, C, 32, 32, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
wt, bs = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
ref = ops.batchnorm_act(x, wt, bs, rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2)
got = ops.batchnorm_act(x, wt, bs, rm.clone(), rv.clone(), True, 0.1, 1e-5, 0.2, out_split=True)
a, b = got.view(torch.int32), _to_split_words(ref).view(torch.int32)
bad = torch.nonzero(a != b)
print('norm out_split vs torch split: mismatches', bad.shape[0], 'of', a.numel())
for i in bad[:6]:
    t = tuple(int(v) for v in i)
    print('  ', t, float(ref[t]), hex(int(a[t]) & 0xffffffff), hex(int(b[t]) & 0xffffffff))
