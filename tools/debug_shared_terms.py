"""Debug aid: discriminator gradients of one step, shared vs separate layout-term evaluation, plus the run-to-run noise floor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd.data.args import make_opt
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.data.vocab import cater_vocab
from ag2video_amd.engine import TrainStep
flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
         '--gconv_hidden_dim', '64', '--gpu_ids', '0']
dev = torch.device('cuda:0')
def run(separate):
    opt = make_opt(cater_vocab(), flags)
    opt.no_shared_layout_terms = separate
    ts = TrainStep(opt, dev, weights_seed=5)
    D = ts.discriminator.img_discriminator
    b0 = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
    ts.gd_step(b0, apply_updates=False)
    return {n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None}
order = [bool(int(x)) for x in (sys.argv[1] if len(sys.argv) > 1 else '0011')]
a, b, c, d = [run(m) for m in order]
print('modes (1 = separate):', order)
from oracle.step import OracleTrainer
opt = make_opt(cater_vocab(), flags)
ts = TrainStep(opt, dev, weights_seed=5)
orc = OracleTrainer(opt, {k: list(v.shape) for k, v in ts.model.state_dict().items()},
                    {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}, 5)
orc.gd_step(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11), apply_updates=False)
ref = {k: v.to(dev) for k, v in orc.last_D_grads.items() if v is not None}
def cmpref(x, tag):
    ks = [k for k in ref if k in x]
    worst = sorted(((x[k] - ref[k]).abs().max().item() / max(1e-12, ref[k].abs().max().item()), k) for k in ks)[-3:]
    print(tag, len(ks), ['%s %.2e' % (k, e) for e, k in worst])
for t, g in zip('abcd', (a, b, c, d)):
    cmpref(g, 'run %s vs oracle' % t)

def cmp(x, y, tag):
    worst = sorted(((x[k] - y[k]).abs().max().item() / max(1e-12, y[k].abs().max().item()), k) for k in y)[-4:]
    print(tag, ['%s %.2e' % (k, e) for e, k in worst])
cmp(a, b, 'a vs b')
cmp(c, d, 'c vs d')
cmp(b, c, 'b vs c')
w = 'discriminator_0.model0.0.weight'
e = (a[w] - c[w]).abs()
print('model0.0.weight: image channels', e[:, :3].max().item(), 'layout channels', e[:, 3:].max().item(), 'max', c[w].abs().max().item())
