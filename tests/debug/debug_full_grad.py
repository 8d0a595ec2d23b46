"""Where does a full gradient tensor of a default-width step differ from the fp64 oracle?  (tests/debug: uses the oracle.)
usage: python tests/debug/debug_full_grad.py c2p discriminator_0.model3.0.0.weight_orig [fp32|bf16x3]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import _cases as C  # noqa: E402


def main():
    case, name = sys.argv[1], sys.argv[2]
    prec = sys.argv[3] if len(sys.argv) > 3 else 'bf16x3'
    from ag2video_amd import ops
    from ag2video_amd.engine import TrainStep
    dev = torch.device('cuda:0')
    ops.set_conv_precision(prec)
    opt, side, frames, batch = C.case_opt(case)
    donly = len(sys.argv) > 4 and sys.argv[4] == 'donly'
    with ops.deterministic():
        ts = TrainStep(opt, dev, weights_seed=C.WEIGHTS_SEED)
        sG = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
        sD = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
        clip = C.case_clip(case, opt, batch, frames, side, C.CLIP_SEED)
        if not donly:
            ts.gd_step(ts.to_device(clip), apply_updates=False)
    tag = 'D_grads' if name.startswith('discriminator') or name in sD else 'G_grads'
    C.FULL_GRADS = {tag: (name,)}
    torch.set_num_threads(16)
    o64 = C.oracle_step(case, sG, sD, None, torch.float64)
    o32 = C.oracle_step(case, sG, sD, None, torch.float32)
    if donly:
        # the discriminator loss on the ORACLE's generated frames: is the difference in D's kernels, or in D's input (the GPU's imgs_pred
        # differs from o64's by ~4e-4 of max)?
        with ops.deterministic():
            b = ts.to_device(clip)
            imgs, objs, boxes, triplets, actions, _ = b
            mo = ts.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
            ts.gans_model(b, mo, mode="compute_generator_loss")          # (two D passes: the power iterations advance as in gd_step)
            print('imgs_pred GPU vs o64: max err / max', float((mo[0].detach().double().cpu() - o64['out'][0].double()).abs().max() / o64['out'][0].abs().max()))
            mo2 = (o64['out'][0].float().to(dev),) + tuple(mo[1:])
            DL = ts.gans_model(b, mo2, mode="compute_discriminator_loss")
            DL = {k: v.mean() for k, v in DL.items()}
            ts.discriminator.optimizer_d_img.zero_grad()
            with ops.accumulate_into_param_grads():
                DL["total_img_loss"].backward()
            ops.join_direct_grad_streams()
            torch.cuda.synchronize()
            print('D losses GPU (oracle frames):', {k: float(v) for k, v in DL.items()}, 'o64:', o64['DL'])
    ref = o64['full_grads'][(tag, name)].double()
    r32 = o32['full_grads'][(tag, name)].double()
    mods = dict(ts.model.named_parameters()) if tag == 'G_grads' else dict(ts.discriminator.img_discriminator.named_parameters())
    g = mods[name].grad.detach().double().cpu()
    d = (g - ref)
    print('shape', tuple(g.shape), 'max|ref|', float(ref.abs().max()), 'GPU max err', float(d.abs().max()), 'L2', float(d.norm() / ref.norm()),
          '| o32: max err', float((r32 - ref).abs().max()), 'L2', float((r32 - ref).norm() / ref.norm()))
    print('abs-sum GPU / o64 / o32:', float(g.abs().sum()), float(ref.abs().sum()), float(r32.abs().sum()))
    if g.dim() == 4:
        e = d.abs()
        print('per tap max err:\n', e.amax(dim=(0, 1)).numpy())
        print('per tap L2 err / L2 ref:\n', (d.pow(2).sum(dim=(0, 1)).sqrt() / ref.pow(2).sum(dim=(0, 1)).sqrt()).numpy())
        ek = d.pow(2).sum(dim=(1, 2, 3)).sqrt() / ref.pow(2).sum(dim=(1, 2, 3)).sqrt().clamp(min=1e-30)
        print('per output channel rel L2: max %.3e median %.3e; worst rows' % (float(ek.max()), float(ek.median())), ek.topk(8).indices.tolist(), ['%.2e' % v for v in ek.topk(8).values.tolist()])
        ec = d.pow(2).sum(dim=(0, 2, 3)).sqrt() / ref.pow(2).sum(dim=(0, 2, 3)).sqrt().clamp(min=1e-30)
        print('per input channel rel L2: max %.3e median %.3e; worst cols' % (float(ec.max()), float(ec.median())), ec.topk(8).indices.tolist())
        # is the error a multiple of the reference (a scale), or of something rank-one (spectral norm's u v^T term)?
        a = float((g * ref).sum() / (ref * ref).sum())
        print('best scale g ~ a * ref: a = %.6f, residual L2 after scaling %.3e' % (a, float((g - a * ref).norm() / ref.norm())))
        D2 = d.reshape(d.shape[0], -1)
        sv = torch.linalg.svdvals(D2)
        print('singular values of the error (top 5) relative to its Frobenius norm:', (sv[:5] / D2.norm()).numpy())


if __name__ == '__main__':
    main()
