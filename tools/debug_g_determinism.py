"""Debug aid: run-to-run determinism of freshly built TrainSteps (same seeds): state_dict, generator outputs, losses, gradients."""
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
ref = None
for i in range(8):
    opt = make_opt(cater_vocab(), flags)
    opt.no_shared_layout_terms = True
    ts = TrainStep(opt, dev, weights_seed=5)
    D = ts.discriminator.img_discriminator
    cur = {'sdG:' + k: v.detach().clone() for k, v in ts.model.state_dict().items()}
    cur.update({'sdD:' + k: v.detach().clone() for k, v in D.state_dict().items()})
    b0 = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
    for j, t in enumerate(b0[:5]):
        cur['batch%d' % j] = t.detach().clone().float()
    mo, GL, DL = ts.gd_step(b0, apply_updates=False)
    for j, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred', 'conf_pred')):
        if torch.is_tensor(mo[j]):
            cur['out:' + name] = mo[j].detach().clone()
    cur.update({'GL:' + k: v.detach().clone() for k, v in GL.items()})
    cur.update({'DL:' + k: v.detach().clone() for k, v in DL.items()})
    cur.update({'gradG:' + n: p.grad.detach().clone() for n, p in ts.model.named_parameters() if p.grad is not None})
    cur.update({'gradD:' + n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
        print('keys', len(cur))
        continue
    bad = [(k, ((cur[k].float() - ref[k].float()).abs().max() / ref[k].float().abs().max().clamp_min(1e-12)).item()) for k in ref]
    bad = [(k, e) for k, e in bad if e > 2e-4]
    print('run', i, 'differs from run 0 in', len(bad), [('%s %.1e' % (k, e)) for k, e in bad[:10]])
