"""Micro-benchmark of the implicit-GEMM conv kernels at c3 (256x256, B=8) layer shapes. GPU box only."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import ops, _lib
from ag2video_amd.ops import _ptr, _stream

SHAPES = [  # name, N, C, H, W, K, R, stride, pad
    ('conv_dim_in', 8, 1027, 256, 256, 512, 3, 1, 1),
    ('spade_shared_256', 8, 512, 256, 256, 128, 3, 1, 1),
    ('spade_gamma_256_c64', 8, 128, 256, 256, 64, 3, 1, 1),
    ('up3_conv0', 8, 128, 256, 256, 64, 3, 1, 1),
    ('up2_conv0', 8, 256, 128, 128, 128, 3, 1, 1),
    ('head_conv', 8, 1024, 8, 8, 1024, 3, 1, 1),
    ('mid_conv_16', 8, 1024, 16, 16, 1024, 3, 1, 1),
    ('d_model0', 32, 259, 256, 256, 64, 4, 2, 2),
    ('d_model1', 32, 64, 129, 129, 128, 4, 2, 2),
    ('conv_img', 8, 64, 256, 256, 3, 3, 1, 1),
    ('flow_res_32', 8, 256, 32, 32, 256, 3, 1, 1),
    ('spade_shared_64', 8, 512, 64, 64, 128, 3, 1, 1),
    ('g_mid_16_1024_128', 8, 128, 16, 16, 1024, 3, 1, 1),
    # frame-batched chunk shapes (round 2)
    ('b_shared_fwd', 31, 512, 256, 256, 128, 3, 1, 1),
    ('b_merged_384', 24, 512, 256, 256, 384, 3, 1, 1),
    ('b_gamma_beta', 40, 128, 256, 256, 128, 3, 1, 1),
    ('b_mid_16', 120, 1024, 16, 16, 1024, 3, 1, 1),
    ('b_up2', 120, 256, 128, 128, 128, 3, 1, 1),
]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if s[0] in sys.argv[1:]]
dev = torch.device('cuda:0')
res = []
for name, N, C, H, W, K, R, st, pad in SHAPES:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, R, R, device=dev) * (C * R * R) ** -0.5
    OH = (H + 2 * pad - R) // st + 1
    y = torch.empty(N, K, OH, OH, device=dev)
    dy = torch.randn(N, K, OH, OH, device=dev)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    wsb = torch.empty(max(1, _lib.query('ag2v_conv2d_workspace_floats', C, K, R, R)), device=dev)
    flops = 2.0 * N * K * OH * OH * C * R * R
    def run(which):
        s = _stream()
        if which == 'fwd':
            _lib.call('ag2v_conv2d_fwd', _ptr(x), _ptr(w), None, None, _ptr(y), N, C, H, W, K, R, R, st, pad, _ptr(wsb), s)
        elif which == 'dgrad':
            _lib.call('ag2v_conv2d_dgrad', _ptr(dy), _ptr(w), _ptr(dx), N, C, H, W, K, R, R, st, pad, _ptr(wsb), s)
        else:
            _lib.call('ag2v_conv2d_wgrad', _ptr(dy), _ptr(x), _ptr(dw), N, C, H, W, K, R, R, st, pad, s)
    for which in ('fwd', 'dgrad', 'wgrad'):
        run(which); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get('REPS', '10'))
        e0.record()
        for _ in range(reps): run(which)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        r = dict(layer=name, op=which, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))
        print(json.dumps(r), flush=True)
        res.append(r)
    del x, w, y, dy, dx, dw
    torch.cuda.empty_cache()
