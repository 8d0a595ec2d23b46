"""conv_wgrad_halo_bf16x3_kernel against the 16-byte-load weight-gradient kernels it replaces, on c3 layer shapes (frame-batched chunk
sizes), alternating in one process: time / TF-equivalent, and the difference of the results.  usage: wgrad_halo_bench.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ag2video_amd import ops, _lib

SHAPES = [(21, 512, 256, 256, 384), (60, 128, 256, 256, 256), (120, 128, 128, 128, 512), (120, 512, 128, 128, 384), (60, 128, 256, 256, 128),
          (120, 128, 64, 64, 1024), (120, 512, 64, 64, 384), (120, 256, 128, 128, 128)]


def to_split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return ((hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)).view(torch.float32)


def mode(new):
    if new:
        os.environ.pop('AG2V_NO_WGRAD_HALO', None)
    else:
        os.environ['AG2V_NO_WGRAD_HALO'] = '1'
    torch.cuda.synchronize()
    ops.reload_kernel_switches()


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get('WGH_SHAPES'):          # e.g. WGH_SHAPES=0,1: a subset (counter runs)
    SHAPES = [SHAPES[int(i)] for i in os.environ['WGH_SHAPES'].split(',')]
ONLY_NEW = os.environ.get('WGH_ONLY_NEW') is not None
if os.environ.get('WGH_CUSTOM'):          # e.g. WGH_CUSTOM='64,512,256,256,384;40,512,128,128,384'
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['WGH_CUSTOM'].split(';')]
dev = torch.device('cuda')
for shp in SHAPES:
    N, C, H, W, K = shp[:5]
    G = shp[5] if len(shp) > 5 else 1          # sample groups (per-frame-group gradients)
    fmt = shp[6] if len(shp) > 6 else 3        # bit 0: x as split words, bit 1: dy as split words
    xs = torch.randn(N, C, H, W, device=dev)
    dys = torch.randn(N, K, H, W, device=dev)
    if fmt & 1:
        xs = to_split(xs)
    if fmt & 2:
        dys = to_split(dys)
    gs = (G,) if G > 1 else ()
    dw = {True: torch.empty(gs + (K, C, 3, 3), device=dev), False: torch.empty(gs + (K, C, 3, 3), device=dev)}
    best, names = {True: 1e9, False: 1e9}, {}
    with ops.conv_precision('bf16x3'), torch.no_grad(), ops.deterministic(os.environ.get('WGH_DET') is not None):
        for rnd in range(3):
            for new in ((True,) if ONLY_NEW else (False, True)):
                mode(new)
                run = lambda: ops._conv_wgrad_launch(32, dys, xs, dw[new], K, 3, 3, 1, 1, 0, groups=G, in_split=bool(fmt & 1), dy_split=bool(fmt & 2))
                run()
                names[new] = _lib.query('ag2v_last_conv_kernel').decode()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    run()
                b.record()
                torch.cuda.synchronize()
                best[new] = min(best[new], a.elapsed_time(b) / reps)
    mode(True)
    fl = 2.0 * N * K * H * W * C * 9
    if ONLY_NEW:
        dw[False], best[False], names[False] = dw[True], best[True], names[True]
    err = float((dw[True] - dw[False]).abs().max() / dw[False].abs().max())
    print('wgrad %-28s old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF (%+.1f %%) | new-vs-old %.1e | %s | %s'
          % (tuple(shp), best[False], fl / best[False] / 1e9, best[True], fl / best[True] / 1e9, (best[False] / best[True] - 1) * 100, err,
             names[True], names[False].split('<')[0]), flush=True)
    del xs, dys
    torch.cuda.empty_cache()
