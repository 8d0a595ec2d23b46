"""TEST-ONLY: the default-width step cases compared with the oracle (tests/test_configs_gpu.py) and measured for their fp32-vs-fp64
noise floor (tests/golden/make_noise_floor.py -> tests/golden/noise_floor.json).  One definition for both, so the floor that gates
a case was measured on exactly the clip, weights and options the GPU test runs."""
import numpy as np
import torch

# name: (image side, frames, batch).  c1 is BASELINE.json's configs[0]; c2p / c3p are the resolutions of c2 / c3 at the smallest clip
# that keeps their structure (batch 1; 3 / 4 frames: two / three generated frames, i.e. T > 1 frame groups): the 128^2 / 256^2 layer
# stack -- up_2 / up_3, the full-resolution flow network and warp, the discriminators at 129^2 / 65^2, the 6-level segmentation pyramid.
# c5p / c5pT: c5's geometry -- Something-Else vocabulary (ONE attribute => 128 layout channels, 2-3 objects with the hand last, 9
# predicates / 9 action classes: reference data/smth.py:85-104), 224 x 224 = 7 * 32 (the non-power-of-two pyramid 7, 14, ..., 224 of
# spade_generator.py:53-81; discriminator maps 113^2 / 57^2); c5pT adds --temporal_D 1 --frames_per_action 4 (the reference's
# MultiscaleDiscriminatorT, discriminator.py:16-59, over one window of 4 generated frames: 5 frames).
# c1bp: c1 TRAINED with --bp_prev 1 (reference generator.py:68-71: the generated previous frame is fed back, so the loss of frame t
# back-propagates through frames t-1, t-2: the sequential generator path under autograd).
STEP_CASES = {'c1': (64, 4, 1), 'c2p': (128, 3, 1), 'c3p': (256, 4, 1), 'c5p': (224, 3, 1), 'c5pT': (224, 5, 1), 'c1bp': (64, 4, 1)}
# FULL-size cases whose oracle runs are committed as fixtures (tests/golden/make_oracle_c2.py -> oracle_c2.npz) instead of running
# inside a test: c2 = BASELINE.json's configs[1], CATER 128 x 128, 8 frames, batch 4.
# c3h: the benchmarked resolution with two samples per batch-norm statistic (256 x 256, 4 frames, batch 2).
# c5h: c5's geometry (Something-Else vocabulary, 224 x 224) at batch 2, 4 frames.
FULL_CASES = {'c2': (128, 8, 4), 'c3h': (256, 4, 2), 'c5h': (224, 4, 2)}
WEIGHTS_SEED, CLIP_SEED, GRAPH_CLIP_SEED = 1, 1234, 4321


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def case_opt(case):
    import bench
    side, frames, batch = (STEP_CASES.get(case) or FULL_CASES[case])
    if case.startswith('c5'):
        from ag2video_amd.data.args import make_opt
        from ag2video_amd.data.vocab import smth_vocab
        flags = ['--no_vgg_loss', '--image_size', '%d,%d' % (side, side), '--batch_size', str(batch), '--gpu_ids', '0', '--dataset', 'smth_else']
        if case.endswith('T'):
            flags += ['--temporal_D', '1', '--frames_per_action', '4']
        return make_opt(smth_vocab(), flags), side, frames, batch
    opt, _, _, _ = bench.make_opt('c1')
    opt.image_size = (side, side)
    opt.batch_size = batch
    opt.gpu_ids = [0]
    if case.endswith('bp'):
        opt.bp_prev = 1
    return opt, side, frames, batch


def case_clip(case, opt, batch, frames, side, seed):
    from ag2video_amd.data.synthetic import make_clip_batch
    if case.startswith('c5'):        # bench.make_clip's Something-Else shape: 2-3 objects (the hand last), <= 2 actions
        return make_clip_batch(opt.vocab, batch, frames, (side, side), seed=seed, min_objs=2, max_objs=3, max_actions=2)
    return make_clip_batch(opt.vocab, batch, frames, (side, side), seed=seed)


# Full gradient TENSORS kept by oracle_step beside the statistics and compared element-wise by tests/test_configs_gpu.py (VERDICT r05
# next #6c): the decoder's last 3x3 convolution, the 1027 -> 512 layout convolution, the discriminator's widest 4x4 layer.
FULL_GRADS = {'G_grads': ('layout_to_video.module.netG.up_3.conv_1.weight_orig', 'layout_to_video.module.conv_dim_in.0.0.weight_orig'),
              'D_grads': ('discriminator_0.model3.0.0.weight_orig',)}


def is_front(name):
    """Tensors in FRONT of the whole generator (flow network, appearance graph net): their gradient sums thousands of decoder paths."""
    return 'flows_network' in name or name.startswith('acts_to_objs')


def oracle_step(case, shapes_G, shapes_D, shapes_T=None, dtype=torch.float32, keep_grads=False):
    """One oracle iteration (G step, D step, graph step) of a case, kept as what the checks need: outputs, losses, per-parameter
    gradient statistics (or the gradients themselves), post-step state statistics + buffers."""
    from oracle.step import OracleTrainer
    opt, side, frames, batch = case_opt(case)
    orc = OracleTrainer(opt, shapes_G, shapes_D, WEIGHTS_SEED, key_shapes_T=shapes_T, dtype=dtype)
    clip = case_clip(case, opt, batch, frames, side, CLIP_SEED)
    mo_r, GL_r, DL_r = orc.gd_step(clip)
    keep = (lambda g: g.detach().clone()) if keep_grads else stats
    res = {'out': [mo_r[i].detach().clone() for i in range(4)],
           'GL': {k: float(v) for k, v in GL_r.items()}, 'DL': {k: float(v) for k, v in DL_r.items()},
           'G_grads': {n: (None if g is None else keep(g)) for n, g in orc.last_G_grads.items()},
           'D_grads': {n: (None if g is None else keep(g)) for n, g in orc.last_D_grads.items()}}
    res['full_grads'] = {(tag, n): getattr(orc, 'last_' + tag)[n].detach().clone() for tag, names in FULL_GRADS.items() for n in names}
    if shapes_T:
        res['T_grads'] = {n: (None if g is None else keep(g)) for n, g in orc.last_T_grads.items()}
    gclip = case_clip(case, opt, batch, 4 * frames, side, GRAPH_CLIP_SEED)
    bp_r, GG_r = orc.graph_step(gclip)
    res['bp'] = bp_r.detach().clone()
    res['GG'] = {k: float(v) for k, v in GG_r.items()}
    state = {}
    for tag, sd in (('G', orc.G), ('D', orc.D)):
        for k, v in sd.items():
            r = v.detach()
            buf = (not v.requires_grad) and r.is_floating_point()
            state[(tag, k)] = (stats(r), r.numel(), r.clone() if buf else None)
    res['state'] = state
    return res


def cpu_key_shapes(case):
    """state_dict key -> shape tables of a case, from the product's modules built on the CPU (no kernels run)."""
    from ag2video_amd.models.meta_models import AG2VideoModel, MetaDiscriminatorModel
    opt, _, _, _ = case_opt(case)
    opt.gpu_ids = []
    cpu = torch.device('cpu')
    sG = {k: list(v.shape) for k, v in AG2VideoModel(opt, cpu).state_dict().items()}
    md = MetaDiscriminatorModel(opt)
    sD = {k: list(v.shape) for k, v in md.img_discriminator.state_dict().items()}
    dt = getattr(md, 'temporal_discriminator', None)
    sT = {k: list(v.shape) for k, v in dt.state_dict().items()} if dt is not None else None
    return sG, sD, sT


def out_errors(a, b):
    """(max-norm error / max|b|, relative L2) of tensor a against reference b, in fp64."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).sum().sqrt() / b.pow(2).sum().sqrt())
