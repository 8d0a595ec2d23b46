"""TIMING EXPERIMENT (results of the patched steps are stale, not computed): what do the graph-net programs cost the c3 step?
The programs run on side streams beside the convolutions (17-21 ms of kernel time per step in rocprofv3's sum); this measures how
much of that the step's wall time actually sees: c3 steps with Program.run() replaced by a no-op (the arena keeps the previous
step's values -- same clip, same weights apart from Adam's update -- so every downstream kernel runs on valid numbers)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from ag2video_amd import graph_program as gp
    from ag2video_amd.engine import TrainStep
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    opt, side, frames, batch = bench.make_opt(cfg)
    dev = torch.device('cuda:0')
    ts = TrainStep(opt, dev, weights_seed=5)
    clip = ts.to_device(bench.make_clip(opt, cfg, batch, frames, side, seed=1234))
    for _ in range(3):
        ts.gd_step(clip)
    real_run = gp.Program.run

    def timed(k=5):
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(k):
            ts.gd_step(clip)
        torch.cuda.synchronize()
        return (time.time() - t0) * 1e3 / k
    for rep in range(3):
        a = timed()
        gp.Program.run = lambda self: None
        b = timed()
        gp.Program.run = real_run
        print('%s step: %.1f ms with the graph-net programs, %.1f ms with their launches skipped (%.1f ms)' % (cfg, a, b, a - b), flush=True)


if __name__ == '__main__':
    main()
