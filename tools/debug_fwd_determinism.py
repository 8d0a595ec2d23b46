"""Debug: toy golden forward (iteration 0) -- imgs_pred error vs golden per precision, and run-to-run determinism."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ag2video_amd import ops
from ag2video_amd.data.args import make_opt
from ag2video_amd.data.vocab import cater_vocab
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep
gold = np.load('tests/golden/toy64_step.npz')
opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags']))); toy = json.loads(str(gold['meta.toy']))
opt.gpu_ids = [0]
dev = torch.device('cuda:0')
outs = {}
for prec in ('fp32', 'bf16x3', 'fp32', 'bf16x3'):
    ops.set_conv_precision(prec)
    ts = TrainStep(opt, dev, weights_seed=toy['wseed'])
    batch = ts.to_device(make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed']))
    mo, GL, DL = ts.gd_step(batch)
    row = []
    for name, got in zip(('imgs_pred', 'boxes_pred', 'flows_pred'), mo[:3]):
        ref = gold[name]; g = got.detach().cpu().numpy()
        row.append('%s %.2e' % (name, np.abs(g - ref).max() / np.abs(ref).max()))
        if (prec, name) in outs:
            row.append('rerun-diff %.2e' % (np.abs(g - outs[(prec, name)]).max() / np.abs(ref).max()))
        outs[(prec, name)] = g
    print(prec, ' | '.join(row), flush=True)
