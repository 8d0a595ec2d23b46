"""Debug aid: repeat the discriminator loss forward/backward on fixed inputs and identical module state; report which
feature maps / gradients differ between repetitions."""
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
opt = make_opt(cater_vocab(), flags)
opt.no_shared_layout_terms = True
ts = TrainStep(opt, dev, weights_seed=5)
D = ts.discriminator.img_discriminator
b0 = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
imgs, objs, boxes, triplets, actions, video_id = b0
with torch.no_grad():
    mo = ts.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
mo = tuple(m.detach() if torch.is_tensor(m) else [x.detach() for x in m] for m in mo)
bufs = {n: b.detach().clone() for n, b in D.named_buffers()}
feats = {}
def hook(name):
    def f(mod, inp, out):
        feats[name] = out.detach().clone() if torch.is_tensor(out) else None
    return f
for n, m in D.named_modules():
    if n.count('.') == 1 and n.startswith('discriminator_'):
        m.register_forward_hook(hook(n))
ref = None
for i in range(10):
    with torch.no_grad():
        for n, b in D.named_buffers():
            b.copy_(bufs[n])
    feats.clear()
    DL = ts.gans_model(b0, mo, mode="compute_discriminator_loss")
    ts.discriminator.optimizer_d_img.zero_grad()
    DL["total_img_loss"].backward()
    cur = {'loss': DL["total_img_loss"].detach().clone()}
    cur.update({'feat:' + k: v for k, v in feats.items() if v is not None})
    cur.update({'grad:' + n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
        print('keys', len(cur))
        continue
    bad = [(k, ((cur[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-12)).item()) for k in ref]
    bad = [(k, e) for k, e in bad if e > 1e-4]
    print('rep', i, 'differs from rep 0 in', len(bad), [('%s %.1e' % (k, e)) for k, e in bad[:12]])
