"""cProfile of the host side of gd_step at c3: which Python frames the ~8k launches of a step spend their issue time in."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, '.')
import bench
from ag2video_amd.engine import TrainStep
opt, side, frames, batch = bench.make_opt('c3')
ts = TrainStep(opt, 'cuda', weights_seed=0)
clip = ts.to_device(bench.make_clip(opt, 'c3', batch, frames, side, seed=1))
for _ in range(2):
    ts.gd_step(clip)
torch.cuda.synchronize()
t = time.time()
pr = cProfile.Profile()
pr.enable()
ts.gd_step(clip)
pr.disable()
host = time.time() - t
torch.cuda.synchronize()
print('host time of the profiled step %.1f ms (GPU step ~860 ms)' % (host * 1e3))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
