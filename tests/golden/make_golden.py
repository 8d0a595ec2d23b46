"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the *unmodified* reference modules from /root/reference on CPU with the five shims
of SURVEY.md 8(c), fills them with the closed-form weights of ag2video_amd.weights, runs two
full training iterations (G step, D step, graph step: scripts/train.py:440-493) on the seeded
synthetic clips of ag2video_amd.data.synthetic and writes small .npz/.json fixtures next to
this file.  Only data (inputs are re-generated from the seed; outputs, losses, gradient /
parameter checksums) is stored -- never reference source.

    python tests/golden/make_golden.py            # writes tests/golden/toy64_*.npz / *.json
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

TOY_FLAGS = ['--gpu_ids', '-1', '--use_cuda', '0', '--no_vgg_loss', '--image_size', '64,64',
             '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8', '--gconv_hidden_dim', '64']
TOY = dict(batch=2, frames=3, graph_frames=6, size=(64, 64), seed=1234, wseed=7)


def install_shims():
    m = types.ModuleType('torch.tensor')
    m.Tensor = torch.Tensor
    sys.modules['torch.tensor'] = m
    tv = types.ModuleType('torchvision')
    sys.modules['torchvision'] = tv
    for sub in ['models', 'models.inception', 'transforms', 'transforms.functional', 'utils']:
        mm = types.ModuleType('torchvision.' + sub)
        sys.modules['torchvision.' + sub] = mm
        parent = tv if '.' not in sub else sys.modules['torchvision.' + sub.split('.')[0]]
        setattr(parent, sub.split('.')[-1], mm)
    T = sys.modules['torchvision.transforms']
    for n in ['Compose', 'Normalize', 'Resize', 'CenterCrop', 'ToTensor', 'ToPILImage', 'Lambda']:
        setattr(T, n, type(n, (object,), {'__init__': lambda s, *a, **k: None, '__call__': lambda s, x: x}))
    import PIL.Image  # noqa: F401
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.ByteTensor = torch.ByteTensor
    sys.path.insert(0, '/root/reference')


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], np.float64)


def main():
    torch.set_num_threads(8)
    install_shims()
    sys.argv = ['make_golden']
    from data.args import parser, init_args  # reference parser
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.weights import closed_form_fill_
    args = parser.parse_args(TOY_FLAGS)
    args.vocab = cater_vocab()
    init_args(args)
    from models.meta_models import AG2VideoModel, MetaDiscriminatorModel
    from models.spade_models.loss_model import LossModel
    from models.spade_models.networks.sync_batchnorm import DataParallelWithCallback
    dev = torch.device('cpu')
    model = AG2VideoModel(args, dev)
    disc = MetaDiscriminatorModel(args)
    gans = LossModel(args, discriminator=disc)
    gans = DataParallelWithCallback(gans, device_ids=args.gpu_ids).to(dev)
    closed_form_fill_(model, seed=TOY['wseed'], prefix='G.')
    closed_form_fill_(disc.img_discriminator, seed=TOY['wseed'], prefix='D.')
    keys = {
        'model': {k: list(v.shape) for k, v in model.state_dict().items()},
        'd_img': {k: list(v.shape) for k, v in disc.img_discriminator.state_dict().items()},
        'gans_model': {k: list(v.shape) for k, v in gans.state_dict().items()},
    }
    json.dump(keys, open(os.path.join(HERE, 'toy64_state_keys.json'), 'w'), indent=0, sort_keys=True)

    opt_graph = torch.optim.Adam(model.acts_to_boxes.parameters(), lr=args.learning_rate,
                                 betas=(args.beta1, 0.999))
    opt_gen = torch.optim.Adam(list(set(model.parameters()) - set(model.acts_to_boxes.parameters())),
                               lr=args.learning_rate, betas=(args.beta1, 0.999))
    for m in [model, disc]:
        m.train()

    out = {}
    captured = {}

    def cap(name):
        def hook(mod, inp, outp):
            captured.setdefault(name, []).append(outp)
        return hook
    L2V = model.layout_to_video.module
    hooks = [L2V.flows_network.register_forward_hook(cap('flows_network')),
             L2V.conv_dim_in.register_forward_hook(cap('conv_dim_in')),
             L2V.netG.conv_img.register_forward_hook(cap('conv_img')),
             L2V.netG.head_0.register_forward_hook(cap('head_0')),
             L2V.netG.up_3.register_forward_hook(cap('up_3')),
             model.acts_to_objs.register_forward_hook(cap('acts_to_objs'))]

    for it in range(2):
        batch = make_clip_batch(args.vocab, TOY['batch'], TOY['frames'], TOY['size'], seed=TOY['seed'] + it)
        imgs, objs, boxes, triplets, actions, video_id = batch
        captured.clear()
        model_out = model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
        G_losses = gans(batch, model_out, mode="compute_generator_loss")
        G_losses = {k: v.mean() for k, v in G_losses.items()}
        opt_gen.zero_grad()
        G_losses['total_loss'].backward()
        p = 'it%d.' % it
        for k, v in G_losses.items():
            out[p + 'G_loss.' + k] = np.float64(v.item())
        if it == 0:
            imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data = model_out
            out['imgs_pred'] = imgs_pred.detach().numpy()
            out['boxes_pred'] = boxes_pred.detach().numpy()
            out['flows_pred'] = flows_pred.detach().numpy()
            out['conf_pred'] = conf_pred.detach().numpy()
            out['temporal_triplets'] = actions_data[1].numpy()
            out['relative_timesteps'] = actions_data[2].numpy()
            out['objs_obj_vecs'] = captured['acts_to_objs'][0][0].detach().numpy()
            out['flow_t1'] = captured['flows_network'][0][1].detach().numpy()
            out['conv_dim_in_t1.stats'] = stats(captured['conv_dim_in'][0])
            out['head_0_t1.stats'] = stats(captured['head_0'][0])
            out['up_3_t1.stats'] = stats(captured['up_3'][0])
            out['conv_img_t1'] = captured['conv_img'][0].detach().numpy()
            out['conv_img_t2.stats'] = stats(captured['conv_img'][1])
            # generator-side layout of sample 0, frame 1: checksum + an 8-channel slice
            from models.layout import boxes_to_layout
            from models.utils import remove_dummy_objects
            att = L2V.attribute_embedding.forward(objs)
            mask = remove_dummy_objects(objs[0], args.vocab)
            v = torch.cat([att[0][mask], captured['acts_to_objs'][0][0][0][1][mask]], dim=1)
            seg = boxes_to_layout(v, boxes[0][1][mask], 64, 64)
            out['seg_b0_t1.stats'] = stats(seg)
            out['seg_b0_t1.ch0_8'] = seg[0, :8].detach().numpy()
        gstats = {}
        for k, prm in model.named_parameters():
            gstats[k] = None if prm.grad is None else stats(prm.grad)
        if it == 0:
            for k, s in gstats.items():
                if s is not None:
                    out['G_grad.' + k] = s
            out['G_grad_full.netG.conv_img.weight'] = L2V.netG.conv_img.weight.grad.numpy().copy()
            out['G_grad_full.up_3.conv_1.weight_orig'] = L2V.netG.up_3.conv_1.weight_orig.grad.numpy().copy()
            out['G_grad_full.acts_to_objs.acts_embeddings'] = \
                model.acts_to_objs.module.acts_embeddings.weight.grad.numpy().copy()
            out['G_grad_full.acts_to_objs.gconvs.0.net1.0.bias'] = \
                model.acts_to_objs.module.gconvs[0].net1[0].bias.grad.numpy().copy()
            out['G_grad_full.flows.conv_flow.weight'] = L2V.flows_network.conv_flow[0].weight.grad.numpy().copy()
        opt_gen.step()

        D_losses = gans(batch, model_out, mode="compute_discriminator_loss")
        D_losses = {k: v.mean() for k, v in D_losses.items()}
        disc.optimizer_d_img.zero_grad()
        D_losses['total_img_loss'].backward()
        for k, v in D_losses.items():
            out[p + 'D_loss.' + k] = np.float64(v.item())
        if it == 0:
            for k, prm in disc.img_discriminator.named_parameters():
                if prm.grad is not None:
                    out['D_grad.' + k] = stats(prm.grad)
            out['D_grad_full.discriminator_1.model4.0.weight'] = \
                disc.img_discriminator.discriminator_1.model4[0].weight.grad.numpy().copy()
        disc.optimizer_d_img.step()

        gbatch = make_clip_batch(args.vocab, TOY['batch'], TOY['graph_frames'], TOY['size'],
                                 seed=TOY['seed'] + 100 + it)
        gi, go, gb, gt, ga, _ = gbatch
        boxes_pred_g = model(gi, go, gt, ga, boxes_gt=gb, test_mode=False, graph_only=True)
        GG = gans(gbatch, boxes_pred_g, mode="compute_graph_loss")
        GG = {k: v.mean() for k, v in GG.items()}
        opt_graph.zero_grad()
        GG['total_loss'].backward()
        for k, v in GG.items():
            out[p + 'graph_loss.' + k] = np.float64(v.item())
        if it == 0:
            out['graph.boxes_pred'] = boxes_pred_g.detach().numpy()
            for k, prm in model.acts_to_boxes.named_parameters():
                if prm.grad is not None:
                    out['graph_grad.' + k] = stats(prm.grad)
        opt_graph.step()
        for k, v in model.state_dict().items():
            out[p + 'G_state.' + k] = stats(v.float())
        for k, v in disc.img_discriminator.state_dict().items():
            out[p + 'D_state.' + k] = stats(v.float())
    for h in hooks:
        h.remove()

    # discriminator features on fixed inputs after the two steps (eval of the D path alone, train mode)
    batch = make_clip_batch(args.vocab, TOY['batch'], TOY['frames'], TOY['size'], seed=TOY['seed'])
    imgs, objs, boxes, triplets, actions, _ = batch
    with torch.no_grad():
        mo = model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
        ad = [x[:, 1:] for x in mo[4]]
        feats = disc.img_discriminator(imgs[:, 1:], objs, boxes[:, 1:], ad)
    for i, sc in enumerate(feats):
        for j, f in enumerate(sc):
            out['final.D_feat.%d.%d.stats' % (i, j)] = stats(f)
        out['final.D_feat.%d.4' % i] = sc[4].numpy()
    out['meta.flags'] = np.array(json.dumps(TOY_FLAGS))
    out['meta.toy'] = np.array(json.dumps(TOY))
    np.savez_compressed(os.path.join(HERE, 'toy64_step.npz'), **out)
    print('wrote', len(out), 'arrays')
    for k in sorted(out):
        if 'loss' in k:
            print(k, out[k])


if __name__ == '__main__':
    main()
