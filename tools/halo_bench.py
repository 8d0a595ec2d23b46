"""A/B of the halo-tile LDS-DMA convolution kernels (csrc/conv_halo.hip) against the wave-specialised split-word kernels they replace
(conv_igemm_bf16x3_ps_kernel), in ONE process, alternating: forward and stride-1 data gradient on split-word operands.
usage: halo_bench.py [reps] [shape ...]   shape = N,C,H,W,K      (GPU box only; HALO_BASE=32: "old" = the 32x32x16 halo kernels;
HALO_BASE=narrow: "old" = the 64-row layers on the 256-pixel tile, AG2V_NO_HALO_WIDE)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ag2video_amd import ops, _lib

SHAPES = [(31, 512, 256, 256, 128), (40, 384, 256, 256, 512), (60, 128, 256, 256, 256), (60, 256, 256, 256, 128),
          (120, 512, 128, 128, 128), (120, 128, 128, 128, 512), (60, 128, 256, 256, 64), (60, 64, 256, 256, 64),
          (120, 512, 64, 64, 128), (120, 128, 64, 64, 1024)]


def to_split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    w = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xFFFF)
    return w.view(torch.float32)


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    args = sys.argv[1:]
    reps = int(args[0]) if args and args[0].isdigit() else 10
    shapes = [tuple(int(v) for v in a.split(',')) for a in args if ',' in a] or SHAPES
    dev = torch.device('cuda')
    for (N, C, H, W, K) in shapes:
        torch.manual_seed(1)
        x = torch.randn(N, C, H, W, device=dev)
        xs = ops.mark_split(to_split(x))
        del x
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        bias = torch.randn(K, device=dev)
        flops = 2.0 * N * K * H * W * C * 9
        with ops.conv_precision('bf16x3'), ops.bf16x3_min_rows(32), torch.no_grad():
            def fwd():
                return ops.conv2d(xs, w, bias, 1, 1, relu=True)
            res, names, t = {}, {}, {'old': [], 'new': []}
            for mode in ('old', 'new'):
                os.environ.pop('AG2V_NO_HALO', None)
                if mode == 'old':
                    os.environ[{'32': 'AG2V_HALO_MFMA32', 'narrow': 'AG2V_NO_HALO_WIDE'}.get(os.environ.get('HALO_BASE'), 'AG2V_NO_HALO')] = '1'
                ops.reload_kernel_switches()          # the library caches the switches
                res[mode] = fwd()
                names[mode] = _lib.query('ag2v_last_conv_kernel').decode()
                os.environ.pop('AG2V_HALO_MFMA32', None)
                os.environ.pop('AG2V_NO_HALO_WIDE', None)
            torch.cuda.synchronize()
            err = float((res['new'] - res['old']).abs().max() / res['old'].abs().max())
            # fp64 reference on a corner crop of image 0 and of the last image (padding taps included)
            xf = ((xs[[0, N - 1], :, :34, :34].view(torch.int32) & -65536).view(torch.float32).double()
                  + (xs[[0, N - 1], :, :34, :34].view(torch.int32) << 16).view(torch.float32).double())
            ref = torch.relu(torch.nn.functional.conv2d(xf, w.double(), bias.double(), 1, 1))[:, :, :33, :33].float()
            e64 = float((res['new'][[0, N - 1], :, :33, :33] - ref).abs().max() / ref.abs().max())
            e64o = float((res['old'][[0, N - 1], :, :33, :33] - ref).abs().max() / ref.abs().max())
            del res
            for rnd in range(3):
                for mode in ('old', 'new'):
                    os.environ.pop('AG2V_NO_HALO', None)
                    os.environ.pop('AG2V_HALO_MFMA32', None)
                    os.environ.pop('AG2V_NO_HALO_WIDE', None)
                    if mode == 'old':
                        os.environ[{'32': 'AG2V_HALO_MFMA32', 'narrow': 'AG2V_NO_HALO_WIDE'}.get(os.environ.get('HALO_BASE'), 'AG2V_NO_HALO')] = '1'
                    ops.reload_kernel_switches()
                    fwd()
                    t[mode].append(timed(fwd, reps))
            for k_ in ('AG2V_NO_HALO', 'AG2V_HALO_MFMA32', 'AG2V_NO_HALO_WIDE'):
                os.environ.pop(k_, None)
            ops.reload_kernel_switches()
        mo, mn = min(t['old']), min(t['new'])
        print('fwd %-28s old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF (%+.1f %%) | new-vs-old %.1e, vs fp64: new %.1e old %.1e | %s | %s'
              % ((N, C, H, W, K), mo, flops / mo / 1e9, mn, flops / mn / 1e9, (mo / mn - 1) * 100, err, e64, e64o,
                 names['new'].split('<')[0] + '<' + names['new'].split('<')[1], names['old'].split('<')[1]), flush=True)
        del xs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
