"""Diagnostic: c1 (default widths) GPU-vs-oracle output errors for different arithmetic policies.
usage: python tests/debug/debug_c1_precision.py [flow_min_rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from ag2video_amd import ops
from ag2video_amd.data.synthetic import make_clip_batch
from ag2video_amd.engine import TrainStep
from oracle.step import OracleTrainer

dev = torch.device('cuda:0')
opt, side, frames, batch = bench.make_opt('c1')
opt.gpu_ids = [0]
ts = TrainStep(opt, dev, weights_seed=1)
shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
orc = OracleTrainer(opt, shapes_G, shapes_D, 1)
clip = make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234)
imgs, objs, boxes, triplets, actions = clip[:5]
from oracle import model as M
sd0 = {k: v.clone() for k, v in ts.model.state_dict().items()}
with torch.no_grad():
    ref = M.generator_forward({k: v.clone() for k, v in orc.G.items()}, opt, imgs, objs, triplets, actions, boxes, training=True, use_gt=True)
d = ts.to_device(clip)
def run(tag):
    ts.model.load_state_dict(sd0)
    with torch.no_grad():
        got = ts.model(d[0], d[1], d[3], d[4], boxes_gt=d[2], test_mode=False, use_gt=True)
    out = {}
    for i, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred')):
        r = ref[i].detach().numpy()
        out[name] = float(np.abs(got[i].cpu().numpy() - r).max() / np.abs(r).max())
    print(tag, out, flush=True)
ops.set_conv_precision('bf16x3')
for parts in ('', 'res', 'up', 'res,up'):
    os.environ['AG2V_FLOW_SPLIT_PARTS'] = parts
    for rep in range(2):
        run('bf16x3 flow parts on split tiles: [%s]' % parts)
