"""Debug aid: correlate cache hits of LossModel._layout_terms with gradient outliers; optional no-freeze variant."""
import sys, os, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd.data.args import make_opt
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.data.vocab import cater_vocab
from ag2video_amd.engine import TrainStep
from ag2video_amd.models.spade_models import loss_model as LM
flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
         '--gconv_hidden_dim', '64', '--gpu_ids', '0']
dev = torch.device('cuda:0')
variant = sys.argv[1] if len(sys.argv) > 1 else 'plain'
if variant == 'nofreeze':
    LM._params_frozen = lambda m: contextlib.nullcontext()
orig = LM.LossModel._layout_terms
state = {}
def wrapped(self, objs, r_boxes, r_adata, producer):
    cached = getattr(self, '_terms_cache', None)
    t = orig(self, objs, r_boxes, r_adata, producer)
    if not producer:
        state['hit'] = cached is not None and t is cached[2]
    if variant == 'sync':
        torch.cuda.synchronize()
    return t
LM.LossModel._layout_terms = wrapped
ref = None
for i in range(8):
    opt = make_opt(cater_vocab(), flags)
    if variant == 'separate':
        opt.no_shared_layout_terms = True
    ts = TrainStep(opt, dev, weights_seed=5)
    D = ts.discriminator.img_discriminator
    b0 = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
    ts.gd_step(b0, apply_updates=False)
    g = {n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None}
    if ref is None:
        ref = g
    worst = sorted(((g[k] - ref[k]).abs().max().item() / max(1e-12, ref[k].abs().max().item()), k) for k in ref)[-1]
    print(variant, 'run', i, 'hit', state.get('hit'), 'worst vs run 0: %s %.2e' % (worst[1], worst[0]))
