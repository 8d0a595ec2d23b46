"""GPU-vs-oracle drift after one optimizer step (debug)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ag2video_amd.data.args import make_opt
from ag2video_amd.data.vocab import cater_vocab
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep
from oracle.step import OracleTrainer, toy_key_shapes
gold = np.load('tests/golden/toy64_step.npz')
opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags']))); toy = json.loads(str(gold['meta.toy']))
opt.gpu_ids=[0]
dev = torch.device('cuda:0')
kg, kd, _ = toy_key_shapes()
orc = OracleTrainer(opt, kg, kd, toy['wseed'])
ts = TrainStep(opt, dev, weights_seed=toy['wseed'])
def cmp_state(tag):
    rows=[]
    for n,t in ts.model.state_dict().items():
        a=t.detach().cpu().double(); b=orc.G[n].detach().double()
        d=(a-b).abs().max().item(); s=b.abs().max().item()+1e-12
        rows.append((d/s,d,n))
    for n,t in ts.discriminator.img_discriminator.state_dict().items():
        a=t.detach().cpu().double(); b=orc.D[n].detach().double()
        d=(a-b).abs().max().item(); s=b.abs().max().item()+1e-12
        rows.append((d/s,d,'D.'+n))
    rows.sort(reverse=True)
    print('==',tag)
    for r in rows[:14]: print('  rel %.2e abs %.2e %s'%r)
for it in range(2):
    batch = make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed'] + it)
    print('it',it,'shapes', [tuple(b.shape) for b in batch[:5]])
    mo_r, GL_r, DL_r = orc.gd_step(batch)
    mo, GL, DL = ts.gd_step(ts.to_device(batch))
    print(' G', {k:(round(GL[k].item(),5), round(GL_r[k].item(),5)) for k in GL_r})
    print(' D', {k:(round(DL[k].item(),5), round(DL_r[k].item(),5)) for k in DL_r})
    # gradient comparison for D params
    rows=[]
    for n,p in ts.discriminator.img_discriminator.named_parameters():
        g=orc.last_D_grads.get(n)
        if p.grad is None or g is None: continue
        a=p.grad.detach().cpu().double(); b=g.double()
        rows.append(((a-b).abs().max().item()/(b.abs().max().item()+1e-20), n))
    rows.sort(reverse=True); print(' D grad worst', rows[:5])
    rows=[]
    for n,p in ts.model.named_parameters():
        g=orc.last_G_grads.get(n)
        if p.grad is None or g is None: continue
        a=p.grad.detach().cpu().double(); b=g.double()
        if b.abs().max()<1e-6: continue
        rows.append(((a-b).abs().max().item()/(b.abs().max().item()+1e-20), n))
    rows.sort(reverse=True); print(' G grad worst', rows[:8])
    cmp_state('after it%d gd'%it)
    gb = make_clip_batch(opt.vocab, toy['batch'], toy['graph_frames'], tuple(toy['size']), seed=toy['seed'] + 100 + it)
    _, a = orc.graph_step(gb); _, b = ts.graph_step(ts.to_device(gb))
    print(' graph', b['total_loss'].item(), a['total_loss'].item())
