"""Diagnostic: frame-batched vs sequential generator pass at a bench configuration; per-output, per-frame differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
dev = torch.device('cuda:0')
opt, side, frames, batch = bench.make_opt(cfg)
if len(sys.argv) > 2:
    batch = int(sys.argv[2]); opt.batch_size = batch
if len(sys.argv) > 3:
    frames = int(sys.argv[3])
opt.gpu_ids = [0]
ts = TrainStep(opt, dev, weights_seed=0)
clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
sd0 = {k: v.clone() for k, v in ts.model.state_dict().items()}
outs = []
MODES = {'bs': (False, True), 'ss': (True, True), 'bb': (False, False)}[os.environ.get('PAIR', 'bs')]
for seq in MODES:
    ts.model.load_state_dict(sd0)
    opt.batch_frames = not seq
    with torch.no_grad():
        o = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
    outs.append([t.clone() for t in o[:4]])
for name, a, b in zip(('imgs_pred', 'boxes_pred', 'flows_pred'), outs[0], outs[1]):
    print(name, 'max rel err', ((a - b).abs().max() / b.abs().max()).item(), 'finite', bool(torch.isfinite(a).all()), bool(torch.isfinite(b).all()))
    if a.dim() == 5:
        for f in range(a.shape[1]):
            print('   frame', f, ((a[:, f] - b[:, f]).abs().max() / b.abs().max()).item(), ' per-sample:',
                  [round(((a[n, f] - b[n, f]).abs().max() / b.abs().max()).item(), 5) for n in range(a.shape[0])])
