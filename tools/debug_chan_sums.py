"""Which backward paths still sum an activation-sized split-word gradient per channel in their own pass (ag2v_channel_sum_acc_s),
and which data-gradient launches added the sums in their epilogue (ag2v_conv_opts::chan_sums_done) -- one c3 step.
usage (GPU box): python tools/debug_chan_sums.py"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from ag2video_amd import _lib, ops
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = bench.make_opt('c3')
    dev = torch.device('cuda:0')
    ts = TrainStep(opt, dev, weights_seed=5)
    clip = ts.to_device(bench.make_clip(opt, 'c3', batch, frames, side, seed=1234))
    ts.gd_step(clip)
    torch.cuda.synchronize()
    seen = collections.Counter()
    real_call = _lib.call
    real_dgrad = ops._conv_dgrad_launch

    def call(name, *a):
        if name == 'ag2v_channel_sum_acc_s':
            st = [f for f in traceback.extract_stack() if f.filename.endswith('ops.py')]
            seen[('sum pass', a[3], a[2] * a[4], 'split' if a[6] else 'fp32', 'ops.py:%d' % st[-1].lineno)] += 1
        return real_call(name, *a)

    def dgrad(mr, dy, w, mask, dx, *a, **k):
        done = real_dgrad(mr, dy, w, mask, dx, *a, **k)
        if k.get('chan_sums') is not None or mask is not None:
            seen[('dgrad', tuple(dx.shape), 'K=%d' % w.shape[0], _lib.query('ag2v_last_conv_kernel').decode()[:60], 'gscale' if (len(a) > 3 and a[3] is not None) else '-', 'mask' if mask is not None else '-', 'sums asked' if k.get('chan_sums') is not None else 'no sums',
                  'done' if done else 'not done', 'slot' if not dx.is_contiguous() else 'dense')] += 1
        return done

    _lib.call = call
    ops._conv_dgrad_launch = dgrad
    ts.gd_step(clip)
    torch.cuda.synchronize()
    for k, v in sorted(seen.items(), key=str):
        print(v, k)


if __name__ == '__main__':
    main()
