"""Per-step wall times of the c3 step (host-synchronised per step): the spread, not only the mean -- a sporadic stall of one launch
(a graph-net program left partly resident beside the persistent convolutions) shows as outliers here and only as +1-2 ms in the mean.
usage: step_times.py [steps=30]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from ag2video_amd.engine import TrainStep
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    opt, side, frames, batch = bench.make_opt('c3')
    dev = torch.device('cuda:0')
    ts = TrainStep(opt, dev, weights_seed=5)
    clip = ts.to_device(bench.make_clip(opt, 'c3', batch, frames, side, seed=1234))
    for _ in range(3):
        ts.gd_step(clip)
    torch.cuda.synchronize()
    t = []
    t_all = time.time()
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ts.gd_step(clip)
        e1.record()
        t.append((e0, e1))
    torch.cuda.synchronize()
    wall = (time.time() - t_all) * 1e3 / n
    ms = sorted(a.elapsed_time(b) for a, b in t)
    print('%d steps: mean wall %.1f ms; per step (events) min %.1f  median %.1f  p90 %.1f  max %.1f; steps above median + 5 ms: %d'
          % (n, wall, ms[0], ms[n // 2], ms[int(n * 0.9)], ms[-1], sum(1 for v in ms if v > ms[n // 2] + 5)), flush=True)


if __name__ == '__main__':
    main()
