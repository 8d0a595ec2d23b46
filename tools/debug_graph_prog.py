import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch
import test_graph_program_gpu as T
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.models.graph_models.model import Acts2LayoutModel
from ag2video_amd.weights import closed_form_fill_
dev = torch.device('cuda:0')
opt = T._opt(512)
net = Acts2LayoutModel(opt).to(dev)
closed_form_fill_(net, seed=3, prefix='G.acts_to_objs.module.')
batch, frames = 8, 16
clip = make_clip_batch(opt.vocab, batch, frames, (64, 64), seed=5)
objs, boxes, triplets, actions = [t.to(dev) for t in (clip[1], clip[2], clip[3], clip[4])]
wt = torch.randn(batch, frames, objs.shape[1], 128, generator=torch.Generator().manual_seed(1)).to(dev)
wb = torch.randn(batch, frames, objs.shape[1], 4, generator=torch.Generator().manual_seed(2)).to(dev)
def run():
    net.zero_grad(set_to_none=True)
    tv, bp, adata = net(objs, triplets, actions, boxes)
    ((tv * wt).sum() + (bp * wb).sum()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
g, g_r = T._both(run)
g2 = run()
for n in g_r:
    e = (g[n] - g_r[n]).abs().max().item() / max(g_r[n].abs().max().item(), 1e-9)
    e2 = (g[n] - g2[n]).abs().max().item() / max(g_r[n].abs().max().item(), 1e-9)
    print('%-55s prog vs ops %.2e   prog vs prog %.2e' % (n, e, e2))
def fwd_only():
    with torch.no_grad():
        tv, bp, _ = net(objs, triplets, actions, boxes)
    return tv.double(), bp.double()
(tv, bp), (tv_r, bp_r) = T._both(fwd_only)
print('forward: tvecs rel err %.2e  boxes rel err %.2e' % (((tv - tv_r).abs().max() / tv_r.abs().max()).item(), ((bp - bp_r).abs().max() / bp_r.abs().max()).item()))
for t in range(tv.shape[1]):
    print(t, '%.2e %.2e' % (((tv[:, t] - tv_r[:, t]).abs().max() / tv_r.abs().max()).item(), ((bp[:, t] - bp_r[:, t]).abs().max() / bp_r.abs().max()).item()))
