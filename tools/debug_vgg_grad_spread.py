import json, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ag2video_amd import ops
from ag2video_amd.models.spade_models.networks.loss import VGGLoss
from ag2video_amd.weights import closed_form_fill_
gold = np.load('/root/repo/tests/golden/extras.npz', allow_pickle=True)
ops.set_conv_precision('bf16x3')
vl = VGGLoss([]).to('cuda')
closed_form_fill_(vl, seed=3, prefix='V.')
rg = gold['vgg.grad_x']
res = []
for i in range(40):
    x = torch.from_numpy(gold['vgg.x']).to('cuda').requires_grad_(True)
    y = torch.from_numpy(gold['vgg.y']).to('cuda')
    loss = vl(x, y); loss.backward()
    gg = x.grad.cpu().numpy()
    res.append((np.abs(gg - rg).max() / np.abs(rg).max(), np.abs(gg - rg).mean() / np.abs(rg).mean(), np.linalg.norm(gg - rg) / np.linalg.norm(rg)))
res = np.array(res)
print('max-ratio  min/median/max', res[:, 0].min(), np.median(res[:, 0]), res[:, 0].max())
print('mean-ratio min/median/max', res[:, 1].min(), np.median(res[:, 1]), res[:, 1].max())
print('rel-L2     min/median/max', res[:, 2].min(), np.median(res[:, 2]), res[:, 2].max())
