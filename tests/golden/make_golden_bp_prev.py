"""Golden vectors for a TRAINING step with --bp_prev 1 (runs ONLY in the build container, where /root/reference exists).

With --bp_prev 1 the reference's generator feeds its own GENERATED frame back as the previous frame during training
(models/spade_models/networks/generator.py:68-71), so the loss of frame t back-propagates through frames t-1, t-2, ...: the
teacher-forced golden (make_golden.py, bp_prev 0) never exercises that path.  Same shims, toy widths, closed-form weights and
seeded clip as make_golden.py (3 frames: frame 2 is generated from the GENERATED frame 1 -- one level of feedback; every further level
amplifies fp32 rounding ~10x, and at 4 frames the reference's own fp32 result already sits 3.6e-2 from the exact one: tests/test_step_gpu.py).  Stores data only ->
tests/golden/bp_prev.npz: outputs, generator / discriminator losses, gradient statistics of every parameter, five full gradient
tensors.

    python tests/golden/make_golden_bp_prev.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, TOY_FLAGS, install_shims, stats  # noqa: E402

FLAGS = TOY_FLAGS + ['--bp_prev', '1']
TOY = dict(batch=2, frames=3, size=(64, 64), seed=1234, wseed=7)


def main():
    torch.set_num_threads(8)
    install_shims()
    sys.argv = ['make_golden_bp_prev']
    from data.args import parser, init_args  # reference parser
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.weights import closed_form_fill_
    args = parser.parse_args(FLAGS)
    args.vocab = cater_vocab()
    init_args(args)
    assert args.bp_prev == 1
    from models.meta_models import AG2VideoModel, MetaDiscriminatorModel
    from models.spade_models.loss_model import LossModel
    from models.spade_models.networks.sync_batchnorm import DataParallelWithCallback
    dev = torch.device('cpu')
    model = AG2VideoModel(args, dev)
    disc = MetaDiscriminatorModel(args)
    gans = DataParallelWithCallback(LossModel(args, discriminator=disc), device_ids=args.gpu_ids).to(dev)
    closed_form_fill_(model, seed=TOY['wseed'], prefix='G.')
    closed_form_fill_(disc.img_discriminator, seed=TOY['wseed'], prefix='D.')
    for m in (model, disc):
        m.train()
    out = {}
    batch = make_clip_batch(args.vocab, TOY['batch'], TOY['frames'], TOY['size'], seed=TOY['seed'])
    imgs, objs, boxes, triplets, actions, _ = batch
    model_out = model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
    G_losses = {k: v.mean() for k, v in gans(batch, model_out, mode="compute_generator_loss").items()}
    model.zero_grad()
    G_losses['total_loss'].backward()
    for k, v in G_losses.items():
        out['G_loss.' + k] = np.float64(v.item())
    imgs_pred, boxes_pred, flows_pred, conf_pred, _ = model_out
    out['imgs_pred'] = imgs_pred.detach().numpy()
    out['flows_pred'] = flows_pred.detach().numpy()
    out['conf_pred'] = conf_pred.detach().numpy()
    L2V = model.layout_to_video.module
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            out['G_grad.' + k] = stats(prm.grad)
    out['G_grad_full.netG.conv_img.weight'] = L2V.netG.conv_img.weight.grad.numpy().copy()
    out['G_grad_full.up_3.conv_1.weight_orig'] = L2V.netG.up_3.conv_1.weight_orig.grad.numpy().copy()
    out['G_grad_full.netG.fc.weight'] = L2V.netG.fc.weight.grad.numpy().copy()
    out['G_grad_full.flows.conv_flow.weight'] = L2V.flows_network.conv_flow[0].weight.grad.numpy().copy()
    out['G_grad_full.acts_to_objs.gconvs.0.net1.0.bias'] = model.acts_to_objs.module.gconvs[0].net1[0].bias.grad.numpy().copy()
    D_losses = {k: v.mean() for k, v in gans(batch, model_out, mode="compute_discriminator_loss").items()}
    disc.img_discriminator.zero_grad()
    D_losses['total_img_loss'].backward()
    for k, v in D_losses.items():
        out['D_loss.' + k] = np.float64(v.item())
    for k, prm in disc.img_discriminator.named_parameters():
        if prm.grad is not None:
            out['D_grad.' + k] = stats(prm.grad)
    # the same weights and clip WITHOUT the feedback: how much of the result is the bp_prev path (a test that cannot tell the two
    # modes apart would prove nothing)
    args.bp_prev = 0
    with torch.no_grad():
        mo0 = model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
    out['teacher_forced.imgs_pred_maxdiff'] = np.float64((mo0[0] - imgs_pred.detach()).abs().max().item())
    out['meta.flags'] = np.array(json.dumps(FLAGS))
    out['meta.toy'] = np.array(json.dumps(TOY))
    np.savez_compressed(os.path.join(HERE, 'bp_prev.npz'), **out)
    print('wrote bp_prev.npz:', len(out), 'arrays;', {k: float(v) for k, v in out.items() if k.startswith('G_loss')},
          'imgs_pred differs from the teacher-forced pass by', float(out['teacher_forced.imgs_pred_maxdiff']))


if __name__ == '__main__':
    main()
