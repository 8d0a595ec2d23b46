"""fp64 ground truth (oracle kernels on the CPU through tests/_cpu_ops_shim.py) for the graph net's gradients: which GPU path is right?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_graph_program_gpu as T
from _cpu_ops_shim import oracle_ops
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.models.graph_models.model import Acts2LayoutModel
from ag2video_amd.weights import closed_form_fill_
dev = torch.device('cuda:0')
opt = T._opt(512)
net = Acts2LayoutModel(opt).to(dev)
closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
clip = make_clip_batch(opt.vocab, 8, 16, (64, 64), seed=5)
objs, boxes, triplets, actions = clip[1], clip[2], clip[3], clip[4]
wt = torch.randn(8, 16, objs.shape[1], 128, generator=torch.Generator().manual_seed(1))
wb = torch.randn(8, 16, objs.shape[1], 4, generator=torch.Generator().manual_seed(2))
def run(n, d, dt):
    n.zero_grad(set_to_none=True)
    tv, bp, _ = n(objs.to(d), triplets.to(d), actions.to(d).to(dt), boxes.to(d).to(dt))
    ((tv * wt.to(d).to(dt)).sum() + (bp * wb.to(d).to(dt)).sum()).backward()
    return {k: p.grad.detach().double().cpu() for k, p in n.named_parameters() if p.grad is not None}
g_prog, g_ops = T._both(lambda: run(net, dev, torch.float32))
ref_net = copy.deepcopy(net).cpu().double()
with oracle_ops():
    g_ref = run(ref_net, torch.device('cpu'), torch.float64)
l2 = lambda x, y: max(((x[n] - y[n]).norm() / y[n].norm()).item() for n in y)
print('program   vs fp64 truth: worst rel-L2 %.2e' % l2(g_prog, g_ref))
print('operators vs fp64 truth: worst rel-L2 %.2e' % l2(g_ops, g_ref))
