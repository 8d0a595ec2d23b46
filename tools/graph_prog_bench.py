"""Timing of the graph-net recurrence (Acts2LayoutModel, default width, c3-sized graphs) as a persistent-kernel program and as
per-operator launches: forward, and forward + backward, HIP-event times per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.models.graph_models.model import Acts2LayoutModel
from ag2video_amd.weights import closed_form_fill_
dev = torch.device('cuda:0')
opt, side, frames, batch = bench.make_opt('c3')
net = Acts2LayoutModel(opt).to(dev)
closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
clip = make_clip_batch(opt.vocab, batch, frames, (64, 64), seed=5)
objs, boxes, triplets, actions = [t.to(dev) for t in (clip[1], clip[2], clip[3], clip[4])]


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


def fwd():
    with torch.no_grad():
        net(objs, triplets, actions, boxes)


def fwdbwd():
    net.zero_grad(set_to_none=True)
    tv, bp, _ = net(objs, triplets, actions, boxes)
    (tv.sum() + bp.sum()).backward()


for label, env in (('program', None), ('operators', '1')):
    if env:
        os.environ['AG2V_NO_GRAPH_PROGRAM'] = env
    print('%-10s forward %.3f ms   forward + backward %.3f ms' % (label, timeit(fwd), timeit(fwdbwd)))

# the program launches alone (no host work in between)
from ag2video_amd import graph_program as gp
os.environ.pop('AG2V_NO_GRAPH_PROGRAM', None)
fwdbwd()
torch.cuda.synchronize()
pl = [p for pool in gp._PLANS.values() for p in pool if p.bwd is not None][0]
for grid in (16, 32, 64, 128, 256):
    os.environ['AG2V_GRAPH_GRID'] = str(grid)
    print('grid %3d: forward program (%d ops) %.3f ms   backward program (%d ops) %.3f ms' % (
        grid, pl.fwd.n_ops, timeit(pl.fwd.run, 20), pl.bwd.n_ops, timeit(pl.bwd.run, 20)))
