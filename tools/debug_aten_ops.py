"""Diagnostic: which ATen ops (copy / add / cat / fill) a c3 step still launches, by input shape and Python call site."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep
dev = torch.device('cuda:0')
opt, side, frames, batch = bench.make_opt('c3')
opt.gpu_ids = [0]
ts = TrainStep(opt, dev, weights_seed=0)
clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
ts.gd_step(clip)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.gd_step(clip)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, None])
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::sum', 'aten::contiguous', 'aten::clone'):
        t = e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total
        shp = str(e.input_shapes)[:80]
        stack = [s for s in (e.stack or []) if 'ag2video_amd' in s or 'bench' in s][:2]
        key = (e.name, shp, ' <- '.join(x.split('/')[-1][:70] for x in stack))
        agg[key][0] += 1
        agg[key][1] += t
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.2f ms %5d  %s  %s  | %s' % (v[1] / 1e3, v[0], k[0], k[1], k[2]))
