"""Where do the frame-batched and the sequential generator pass part ways in deterministic mode?  Hooks on the decoder's blocks; prints
the first module whose output differs (c3 sizes by default; SIDE / FRAMES / BATCH env override)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ag2video_amd import ops
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep

dev = torch.device('cuda:0')
opt, side, frames, batch = bench.make_opt('c3')
side = int(os.environ.get('SIDE', side)); frames = int(os.environ.get('FRAMES', 4)); batch = int(os.environ.get('BATCH', 2))
opt.image_size = (side, side); opt.batch_size = batch; opt.gpu_ids = [0]
GRAD = os.environ.get('GRAD') == '1'
with ops.deterministic():
    ts = TrainStep(opt, dev, weights_seed=0)
    clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
    sd0 = {k: v.clone() for k, v in ts.model.state_dict().items()}
    gen = ts.model.layout_to_video.module
    names = ['flows_network', 'conv_dim_in.0.0', 'conv_dim_in.0.1', 'netG.fc', 'netG.head_0', 'netG.G_middle_0', 'netG.G_middle_1', 'netG.up_0',
             'netG.up_1', 'netG.up_2', 'netG.up_3', 'netG.conv_img']
    mods = dict(gen.named_modules())
    rec = {}

    def unsplit(t):
        if ops.is_split(t):
            u = t.view(torch.int32)
            return (u & -65536).view(torch.float32) + (u << 16).view(torch.float32)
        return t

    def hook(name):
        def f(m, i, o):
            o = o[1] if isinstance(o, tuple) else o
            if isinstance(o, dict) or not torch.is_tensor(o):
                return
            rec.setdefault(name, []).append(unsplit(o.detach()).clone())
        return f
    hs = [mods[n].register_forward_hook(hook(n)) for n in names if n in mods]
    outs = {}
    for seq in (False, True):
        ts.model.load_state_dict(sd0)
        opt.batch_frames = not seq
        rec.clear()
        ctx = torch.enable_grad() if GRAD else torch.no_grad()
        with ctx:
            o = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
        outs[seq] = ({k: [t for t in v] for k, v in rec.items()}, o[0].detach().clone())
    T = frames - 1
    for n in names:
        if n not in outs[False][0]:
            print(n, 'not recorded'); continue
        b = outs[False][0][n][0]                       # [T*B, ...] frame-major
        s = torch.cat(outs[True][0][n], dim=0)         # T calls of [B, ...]
        if b.shape != s.shape:
            print(n, 'shapes', tuple(b.shape), tuple(s.shape)); continue
        d = (b - s).abs().max().item() / s.abs().max().item()
        print('%-22s rel diff %.3e  equal=%s' % (n, d, torch.equal(b, s)))
    print('imgs_pred', ((outs[False][1] - outs[True][1]).abs().max() / outs[True][1].abs().max()).item())
