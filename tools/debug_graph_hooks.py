"""Gradients at module boundaries of the per-operator path (box_net output = gbox[s], obj_vecs_net output = dov0[s], gconv outputs)
against the program's stacks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_graph_program_gpu as T
from ag2video_amd import graph_program as gp
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.models.graph_models.model import Acts2LayoutModel
from ag2video_amd.weights import closed_form_fill_
dev = torch.device('cuda:0')
opt = T._opt(512)
net = Acts2LayoutModel(opt).to(dev)
closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
clip = make_clip_batch(opt.vocab, 8, 16, (64, 64), seed=5)
objs, boxes, triplets, actions = [t.to(dev) for t in (clip[1], clip[2], clip[3], clip[4])]
wt = torch.randn(8, 16, objs.shape[1], 128, generator=torch.Generator().manual_seed(1)).to(dev)
wb = torch.randn(8, 16, objs.shape[1], 4, generator=torch.Generator().manual_seed(2)).to(dev)
# program path
tv, bp, _ = net(objs, triplets, actions, boxes)
((tv * wt).sum() + (bp * wb).sum()).backward()
torch.cuda.synchronize()
pl = [p for pool in gp._PLANS.values() for p in pool if p.bwd is not None][0]
G = {k: pl.bwd.tensor(v).clone() for k, v in pl.g.items()}
Fw = {k: pl.fwd.tensor(v).clone() for k, v in pl.f.items()}
# operator path with hooks
os.environ['AG2V_NO_GRAPH_PROGRAM'] = '1'
cap = {'box': [], 'ovn': [], 'g0': [], 'g1': [], 'g2': [], 'fbox': [], 'fovn': []}
def mk(name, fname=None, idx=None):
    def hook(mod, inp, out):
        o = out[idx] if idx is not None else out
        if fname: cap[fname].append(o.detach().clone())
        o.register_hook(lambda g, name=name: cap[name].append(g.detach().clone()))
    return hook
net.box_net.register_forward_hook(mk('box', 'fbox'))
net.obj_vecs_net.register_forward_hook(mk('ovn', 'fovn'))
for l in range(3):
    net.gconvs[l].register_forward_hook(mk('g%d' % l, None, 0))
net.zero_grad(set_to_none=True)
tv, bp, _ = net(objs, triplets, actions, boxes)
((tv * wt).sum() + (bp * wb).sum()).backward()
torch.cuda.synchronize()
S, BO = pl.S, pl.BO
rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
# forward values (call order = step order); gradients arrive in reverse step order
print('forward: obj_vecs_net out', max(rel(Fw['ov0'][s], cap['fovn'][s].reshape(BO, -1)) for s in range(S)))
for s in range(S - 1, -1, -1):
    gi = S - 1 - s
    print('step %2d: d(box_net out) vs gbox %.2e   d(obj_vecs_net out) vs dov0 %.2e   d(gconv2 out) vs dov3 %.2e  d(gconv1 out) vs dov2 %.2e  d(gconv0 out) vs dov1 %.2e' % (
        s, rel(G['gbox'][s], cap['box'][gi].reshape(BO, -1)), rel(G['dov0'][s], cap['ovn'][gi].reshape(BO, -1)),
        rel(G['dov3'][s], cap['g2'][gi].reshape(BO, -1)), rel(G['dov2'][s], cap['g1'][gi].reshape(BO, -1)), rel(G['dov1'][s], cap['g0'][gi].reshape(BO, -1))))
