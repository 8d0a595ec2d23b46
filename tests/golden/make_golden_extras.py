"""Golden vectors for the rows SURVEY.md 8(f) marks "next" + D4 (runs ONLY in the build container, where /root/reference exists).

Imports the *unmodified* reference classes (shims as in make_golden.py, plus stubs for cv2 / skvideo, which the CATER dataset module
imports but the code exercised here never calls) and stores data only -> tests/golden/extras.npz:

  vgg.*       reference VGGLoss / VGG19 (networks/loss.py:102-118, architecture.py:93-123) over a randomly initialised stack of
              torchvision's published vgg19 ``features`` configuration 'E' (torchvision itself is not installed and its ImageNet
              weights need the network: the reference's slicing + loss arithmetic is what is pinned), closed-form weights 'V.*'
  collate.*   reference collate_fn (data/dataset_params.py:8-104) on ragged per-sample items
  cater.*     reference CATERDataset tensor builders (data/cater.py:157-466: objects, triplets, projected boxes, action windows)
              on the synthetic scene of ag2video_amd.data.cater.synthetic_scene(seed)
  dtemp.*     reference MultiscaleDiscriminatorT / NLayerDiscriminatorTM (discriminator.py:16-87: the temporal PatchGAN that stacks
              the clip's frames into channels; BASELINE's "3-D PatchGAN" has no other counterpart in the reference), closed-form 'T.*'
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, install_shims  # noqa: E402

VGG_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def _vgg19_stub(pretrained=False):
    layers, c = [], 3
    for v in VGG_CFG_E:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c = v
    return types.SimpleNamespace(features=nn.Sequential(*layers))


def main():
    torch.set_num_threads(8)
    install_shims()
    sys.modules['torchvision.models'].vgg19 = _vgg19_stub
    for name in ('cv2', 'skvideo', 'skvideo.io'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['skvideo.io'].FFmpegReader = object
    sys.argv = ['make_golden_extras']
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.weights import closed_form_fill_
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---------------------------------------------------------------- VGG perceptual loss
    from models.spade_models.networks.loss import VGGLoss
    vl = VGGLoss([])
    closed_form_fill_(vl, seed=3, prefix='V.')
    x = torch.randn(1, 2, 3, 32, 32, generator=g).requires_grad_(True)
    y = torch.randn(1, 2, 3, 32, 32, generator=g)
    loss = vl(x, y)
    loss.backward()
    out['vgg.x'], out['vgg.y'] = x.detach().numpy(), y.numpy()
    out['vgg.loss'] = np.array(loss.item())
    out['vgg.grad_x'] = x.grad.numpy()
    feats = vl.vgg(x.detach().view(-1, 3, 32, 32))
    for i, f in enumerate(feats):
        out['vgg.feat%d_stats' % i] = np.array([f.sum().item(), f.abs().sum().item(), tuple(f.shape)[1], tuple(f.shape)[2]])
    out['vgg.keys'] = np.array(json.dumps({k: list(v.shape) for k, v in vl.state_dict().items()}))

    # ---------------------------------------------------------------- collate_fn
    from data.dataset_params import collate_fn
    vocab = cater_vocab()
    items = []
    for i, (O, T, A, F) in enumerate([(4, 3, 2, 3), (6, 5, 1, 3), (3, 2, 4, 3)]):
        objs = {a: torch.randint(0, 3, (O,), generator=g) for a in vocab['attributes']}
        items.append((torch.randn(F, 3, 8, 8, generator=g), objs, torch.rand(F, O, 4, generator=g),
                      torch.randint(0, 5, (F, T, 3), generator=g), torch.rand(A, 7, generator=g), 'vid%d' % i))
    items.insert(1, (False, None, None, None, None, None))          # a failed sample is filtered (dataset_params.py:30)
    for i, it in enumerate(items):
        if isinstance(it[0], bool):
            continue
        out['collate.in%d.vid' % i] = it[0].numpy()
        for a, v in it[1].items():
            out['collate.in%d.objs.%s' % (i, a)] = v.numpy()
        out['collate.in%d.boxes' % i], out['collate.in%d.triplets' % i], out['collate.in%d.actions' % i] = it[2].numpy(), it[3].numpy(), it[4].numpy()
    res = collate_fn(vocab, items)
    for n, t in zip(('vids', 'objs', 'boxes', 'triplets', 'actions'), res[:5]):
        out['collate.out.' + n] = t.numpy()
    out['collate.out.ids'] = np.array(json.dumps(res[5]))

    # ---------------------------------------------------------------- CATER tensor builders
    from data.cater import CATERDataset
    from ag2video_amd.data.cater import synthetic_scene
    ds = CATERDataset.__new__(CATERDataset)
    ds.vocab = vocab
    ds.initial_frames_per_sample, ds.frames_per_action = 48, 16
    for seed in (0, 1):
        sg = synthetic_scene(seed, nframes=301)
        p = 'cater.s%d.' % seed
        for a, v in ds.extract_objs(sg).items():
            out[p + 'objs.' + a] = v.numpy()
        boxes = ds.extract_bounding_boxes(sg)
        out[p + 'boxes_all'] = boxes.numpy()
        actions, (s, e) = ds.extract_actions_split(sg, 300, True)
        out[p + 'actions_raw'], out[p + 'window'] = actions.numpy(), np.array([s, e])
        out[p + 'actions_norm'] = ds.normalized_actions(actions, boxes, s, e).numpy()
        frames_lst = list(range(s, e))[0:48:48 // 16]
        out[p + 'triplets'] = ds.extract_triplets(boxes[frames_lst]).numpy()
        out[p + 'frames_lst'] = np.array(frames_lst)
        np.random.seed(11 + seed)                     # the training split draws its window with np.random (cater.py:207)
        actions_t, (s2, e2) = ds.extract_actions_split(sg, 300, False)
        out[p + 'train_window'] = np.array([s2, e2])
        out[p + 'train_actions_norm'] = ds.normalized_actions(actions_t, boxes, s2, e2).numpy()

    # ---------------------------------------------------------------- temporal discriminator (frames stacked into channels)
    from data.args import parser, init_args
    args = parser.parse_args(['--gpu_ids', '-1', '--use_cuda', '0', '--no_vgg_loss', '--image_size', '32,32', '--batch_size', '2',
                              '--ndf', '8', '--frames_per_action', '4'])
    args.vocab = vocab
    init_args(args)
    from models.spade_models.networks.discriminator import MultiscaleDiscriminatorT
    dt = MultiscaleDiscriminatorT(args)
    closed_form_fill_(dt, seed=5, prefix='T.')
    dt.train()
    img = torch.randn(2, 4, 3, 32, 32, generator=g).requires_grad_(True)
    res = dt(img)
    out['dtemp.img'] = img.detach().numpy()
    tot = 0
    for i, scale in enumerate(res):
        for j, f in enumerate(scale):
            out['dtemp.out%d_%d' % (i, j)] = f.detach().numpy()
            tot = tot + (f * torch.linspace(-1, 1, f.numel()).view_as(f)).sum()
    tot.backward()
    out['dtemp.grad_img'] = img.grad.numpy()
    for n, p_ in dt.named_parameters():
        if p_.grad is not None:
            out['dtemp.grad.' + n] = np.array([p_.grad.double().abs().sum().item(), p_.grad.double().pow(2).sum().sqrt().item()])
    out['dtemp.keys'] = np.array(json.dumps({k: list(v.shape) for k, v in dt.state_dict().items()}))
    for k, v in dt.state_dict().items():
        if k.endswith('weight_u'):
            out['dtemp.state.' + k] = v.numpy()
    # the same classes at 64 x 64 with ndf 16 (round 3: the size of the --temporal_D training-step test)
    args64 = parser.parse_args(['--gpu_ids', '-1', '--use_cuda', '0', '--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2',
                                '--ndf', '16', '--frames_per_action', '4'])
    args64.vocab = vocab
    init_args(args64)
    dt64 = MultiscaleDiscriminatorT(args64)
    closed_form_fill_(dt64, seed=7, prefix='T.')
    dt64.train()
    img64 = torch.randn(2, 4, 3, 64, 64, generator=g).requires_grad_(True)
    res64 = dt64(img64)
    out['dtemp64.img'] = img64.detach().numpy()
    tot = 0
    for i, scale in enumerate(res64):
        for j, f in enumerate(scale):
            out['dtemp64.out%d_%d' % (i, j)] = f.detach().numpy()
            tot = tot + (f * torch.linspace(-1, 1, f.numel()).view_as(f)).sum()
    tot.backward()
    out['dtemp64.grad_img'] = img64.grad.numpy()
    for n, p_ in dt64.named_parameters():
        if p_.grad is not None:
            out['dtemp64.grad.' + n] = np.array([p_.grad.double().abs().sum().item(), p_.grad.double().pow(2).sum().sqrt().item()])
    out['dtemp64.keys'] = np.array(json.dumps({k: list(v.shape) for k, v in dt64.state_dict().items()}))
    for k, v in dt64.state_dict().items():
        if k.endswith('weight_u'):
            out['dtemp64.state.' + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'extras.npz'), **out)
    print('wrote extras.npz with %d arrays, %.1f KB' % (len(out), os.path.getsize(os.path.join(HERE, 'extras.npz')) / 1024))


if __name__ == '__main__':
    main()
