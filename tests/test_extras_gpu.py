"""GPU side of the SURVEY 8(f) "next" rows and D4: the product's VGG perceptual loss (conv + ReLU epilogues, the 2x2 max-pool
kernel) and temporal discriminator against golden vectors of the reference classes (tests/golden/extras.npz) and the oracle, and one
full training step with the perceptual loss switched on."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _deterministic_reductions():
    """Every comparison with the reference golden / the oracle in this file runs with ops.deterministic(): the compared value is then
    a property of the build, not of one launch's split-K atomics order (VERDICT r04 next #1b).  The default path's kernels are pinned
    one by one in tests/test_ops_gpu.py, test_halo_gpu.py, test_grad_slots_gpu.py and, as a step, by the spread report of
    tests/test_configs_gpu.py."""
    from ag2video_amd import ops
    with ops.deterministic():
        yield
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(HERE, 'golden', 'extras.npz'), allow_pickle=False)


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (1, 5, 9, 7), (3, 2, 32, 32)], ids=str)
def test_maxpool2(shape, device):
    """Forward, and the backward's routing to the first maximum of each window -- including exact ties (ReLU zeros)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(*shape, generator=g))             # ~half zeros: windows with tied maxima
    x[0, 0, :2, :2] = 0.75                                       # a window of four equal positive values
    xc, xg = x.clone().requires_grad_(True), x.clone().to(device).requires_grad_(True)
    yr, yg = K.maxpool2(xc), ops.maxpool2(xg)
    assert torch.equal(yg.detach().cpu(), yr.detach())
    w = torch.randn(yr.shape, generator=g)
    (yr * w).sum().backward()
    (yg * w.to(device)).sum().backward()
    assert torch.equal(xg.grad.cpu(), xc.grad)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_vgg_loss_matches_reference_golden(prec, gold, device):
    from ag2video_amd import ops
    from ag2video_amd.models.spade_models.networks.loss import VGGLoss
    from ag2video_amd.weights import closed_form_fill_
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        vl = VGGLoss([]).to(device)
        assert {k: list(v.shape) for k, v in vl.state_dict().items()} == json.loads(str(gold['vgg.keys']))
        closed_form_fill_(vl, seed=3, prefix='V.')
        assert not any(p.requires_grad for p in vl.parameters())
        x = torch.from_numpy(gold['vgg.x']).to(device).requires_grad_(True)
        y = torch.from_numpy(gold['vgg.y']).to(device)
        loss = vl(x, y)
        loss.backward()
        ref = float(gold['vgg.loss'])
        assert abs(loss.item() - ref) <= 1e-3 * abs(ref), (loss.item(), ref)
        # d|a - b| is a sign function: a feature pair closer than the arithmetic's rounding flips its whole term.  The reference golden
        # sits 5e-7 from the fp64 oracle here (tests/golden/noise_floor.json `vgg_grad`): fp32 on the CPU flips no term on this input, so
        # every flipped term is this implementation's -- its summation order (fp32 mode: 2e-3 of the largest gradient) or the 2^-17
        # product error of the 3-term bf16 split (2.1e-2 of the largest gradient = one term; mean deviation 3e-4 of the mean).  Single
        # elements are therefore held to 3e-2 / 6e-2 of the largest gradient (one term), the MEAN deviation to 2e-3 and the relative L2
        # to 2e-2, in deterministic mode (the fixture above) where the set of flipped terms is a property of the build
        rg, gg = gold['vgg.grad_x'], x.grad.cpu().numpy()
        tol_max, tol_mean = (3e-2, 2e-3) if prec == 'fp32' else (6e-2, 2e-3)
        assert np.abs(gg - rg).max() <= tol_max * np.abs(rg).max() and np.abs(gg - rg).mean() <= tol_mean * np.abs(rg).mean()
        assert np.linalg.norm(gg - rg) <= 2e-2 * np.linalg.norm(rg)
    finally:
        ops.set_conv_precision(old)


@pytest.mark.parametrize("tag,side,ndf,seed", [('dtemp', 32, 8, 5), ('dtemp64', 64, 16, 7)])
def test_temporal_discriminator_matches_reference_golden(gold, device, tag, side, ndf, seed):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.models.spade_models.networks.discriminator import MultiscaleDiscriminatorT
    from ag2video_amd.weights import closed_form_fill_
    opt = make_opt(cater_vocab(), ['--gpu_ids', '0', '--no_vgg_loss', '--image_size', '%d,%d' % (side, side), '--batch_size', '2',
                                  '--ndf', str(ndf), '--frames_per_action', '4'])
    dt = MultiscaleDiscriminatorT(opt).to(device)
    assert {k: list(v.shape) for k, v in dt.state_dict().items()} == json.loads(str(gold[tag + '.keys']))
    closed_form_fill_(dt, seed=seed, prefix='T.')
    dt.train()
    img = torch.from_numpy(gold[tag + '.img']).to(device).requires_grad_(True)
    res = dt(img)
    tot = 0
    for i, scale in enumerate(res):
        for j, f in enumerate(scale):
            ref = gold[tag + '.out%d_%d' % (i, j)]
            assert np.abs(f.detach().cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), (i, j)
            tot = tot + (f * torch.linspace(-1, 1, f.numel(), device=device).view_as(f)).sum()
    tot.backward()
    ref = gold[tag + '.grad_img']
    assert np.abs(img.grad.cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()
    for k, v in dt.state_dict().items():
        if k.endswith('weight_u'):
            assert np.abs(v.cpu().numpy() - gold[tag + '.state.' + k]).max() <= 1e-4, k
    for n, p in dt.named_parameters():
        key = tag + '.grad.' + n
        if p.grad is not None and key in gold.files:
            got = np.array([p.grad.double().abs().sum().item(), p.grad.double().pow(2).sum().sqrt().item()])
            assert np.allclose(got, gold[key], rtol=2e-2, atol=1e-5), n


def test_train_step_with_temporal_discriminator_matches_oracle(device):
    """--temporal_D 1 (D4 reachable): MetaDiscriminatorModel adds the temporal multiscale PatchGAN, LossModel its hinge
    and feature-matching terms over windows of frames_per_action generated frames, optimizer_d_img owns its parameters.  One full
    iteration against the oracle (which restates the same terms over its golden-pinned temporal_discriminator_forward): loss keys and
    values, gradients reaching the temporal discriminator and -- through it -- the generator, its parameters updated."""
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    from oracle.step import OracleTrainer
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '16', '--nff', '8', '--gconv_hidden_dim', '64',
             '--gpu_ids', '0', '--temporal_D', '1', '--frames_per_action', '4']
    opt = make_opt(cater_vocab(), flags)
    ts = TrainStep(opt, device, weights_seed=6)
    dt = ts.discriminator.temporal_discriminator
    assert ts.gans_model.module.netD_tmp is dt
    assert any(k.startswith('module.netD_tmp.discriminator_0.model0') for k in ts.gans_model.state_dict())
    n_d = len(list(ts.discriminator.img_discriminator.parameters()))
    assert len(ts._d_params) == n_d + len(list(dt.parameters()))
    assert len(ts.discriminator.img_discriminator.state_dict()) == 56                    # the reference's key set is untouched
    shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
    shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
    shapes_T = {k: list(v.shape) for k, v in dt.state_dict().items()}
    orc = OracleTrainer(opt, shapes_G, shapes_D, 6, key_shapes_T=shapes_T)
    clip = make_clip_batch(opt.vocab, 2, 10, (64, 64), seed=21, max_objs=6)          # 9 generated frames: two windows of 4, one frame dropped
    p0 = {n: p.detach().clone() for n, p in dt.named_parameters()}
    _, GL_r, DL_r = orc.gd_step(clip)
    _, GL, DL = ts.gd_step(ts.to_device(clip))
    assert set(GL) == set(GL_r) >= {'GAN_Tmp', 'GAN_Tmp_Feat'} and set(DL) == set(DL_r) >= {'D_tmp_fake', 'D_tmp_real'}
    for got, ref in ((GL, GL_r), (DL, DL_r)):
        for k in ref:
            assert abs(got[k].item() - ref[k].item()) <= 1e-3 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
    bad = []
    for n, p in dt.named_parameters():
        g_ref = orc.last_T_grads.get(n)
        if g_ref is None:
            continue
        got, ref = p.grad.double().abs().sum().item(), g_ref.double().abs().sum().item()
        if abs(got - ref) > 2e-2 * ref + 2e-6 * p.numel():
            bad.append((n, got, ref))
        assert (p.detach() - p0[n]).abs().max().item() > 0, n          # Adam moved it
    assert not bad, bad[:6]


def test_train_step_with_vgg_loss_matches_oracle(device):
    """The full G + D step with the perceptual loss on (--no_vgg_loss absent): loss keys of the reference (loss_model.py:99-100),
    values against the oracle, gradient reaching the generator through VGG."""
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    from oracle.step import OracleTrainer
    flags = ['--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8', '--gconv_hidden_dim', '64',
             '--gpu_ids', '0']
    from ag2video_amd.models.spade_models.loss_model import LossModel
    with pytest.raises(RuntimeError, match='vgg_weights'):       # the default flags must not silently train on a random VGG19
        LossModel(make_opt(cater_vocab(), flags), discriminator=None)
    opt = make_opt(cater_vocab(), flags + ['--vgg_random_init'])
    assert not opt.no_vgg_loss
    ts = TrainStep(opt, device, weights_seed=6)
    sd_gm = ts.gans_model.state_dict()
    assert 'module.criterionVGG.vgg.slice1.0.weight' in sd_gm and len([k for k in sd_gm if 'criterionVGG' in k]) == 26
    shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
    shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
    shapes_V = {k: list(v.shape) for k, v in ts.gans_model.module.criterionVGG.state_dict().items()}
    orc = OracleTrainer(opt, shapes_G, shapes_D, 6, shapes_V)
    clip = make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=17)
    _, GL_r, DL_r = orc.gd_step(clip)
    _, GL, DL = ts.gd_step(ts.to_device(clip))
    assert set(GL) == set(GL_r) == {'GAN_Img', 'GAN_Feat', 'loss_F_Warp', 'VGG', 'total_loss'}
    for got, ref in ((GL, GL_r), (DL, DL_r)):
        for k in ref:
            assert abs(got[k].item() - ref[k].item()) <= 1e-3 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())
    g = ts.model.layout_to_video.module.netG.conv_img.weight.grad
    r = orc.last_G_grads['layout_to_video.module.netG.conv_img.weight']
    assert abs(g.abs().sum().item() - r.abs().sum().item()) <= 2e-2 * r.abs().sum().item()


def test_train_script_on_scene_files(device, tmp_path):
    """python -m ag2video_amd.scripts.train --data_root <CATER-format directory>: scene JSON + cached frames -> loaders -> collate_fn
    -> G/D/graph steps on the GPU -> checkpoint with the reference's keys."""
    from ag2video_amd.data import cater as C
    from ag2video_amd.data.args import get_args
    from ag2video_amd.scripts import train as T
    root = os.path.join(str(tmp_path), 'cater')
    C.write_synthetic_dataset(root, n_videos=3, nframes=60, seed=4, frame_size=(48, 64))
    out = os.path.join(str(tmp_path), 'out')
    args = get_args(['--data_root', root, '--data_nframes', '60', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8',
                     '--nff', '8', '--gconv_hidden_dim', '64', '--no_vgg_loss', '--num_iterations', '2', '--frames_per_action', '3',
                     '--frames_per_action_graph', '1', '--loader_num_workers', '0', '--output_dir', out, '--checkpoint_every', '100',
                     '--print_every', '1'])
    T.main(args)
    ck = torch.load(os.path.join(out, 'latest.pt'), map_location='cpu', weights_only=False)
    assert ck['counters']['t'] == 2 and 'model_state' in ck and json.load(open(os.path.join(out, 'run_args.json')))['data_root'] == root


def test_train_script_on_something_else_files(device, tmp_path):
    """python -m ag2video_amd.scripts.train --dataset smth_else --data_root <Something-Else-format directory>: label table + box tracks +
    frames -> ag2video_amd.data.smth loaders (vocabulary from the dataset's object mapping) -> collate_fn -> G/D/graph steps on the
    GPU -> checkpoint carrying that vocabulary; the inference driver then runs its Something-Else edits from the checkpoint."""
    from ag2video_amd.data import smth as S
    from ag2video_amd.data.args import get_args
    from ag2video_amd.scripts import test as TS
    from ag2video_amd.scripts import train as T
    root, labels = S.write_synthetic_dataset(os.path.join(str(tmp_path), 'smth'), n_videos=6, n_frames=20, frame_size=(64, 48), seed=2)
    out = os.path.join(str(tmp_path), 'out')
    args = get_args(['--dataset', 'smth_else', '--data_root', root, '--train_list', labels, '--image_size', '64,64', '--batch_size', '2',
                     '--ngf', '8', '--ndf', '8', '--nff', '8', '--gconv_hidden_dim', '64', '--no_vgg_loss', '--num_iterations', '2',
                     '--frames_per_action', '4', '--frames_per_action_graph', '2', '--loader_num_workers', '0', '--output_dir', out,
                     '--checkpoint_every', '100', '--print_every', '1'])
    T.main(args)
    ck = torch.load(os.path.join(out, 'latest.pt'), map_location='cpu', weights_only=True)
    assert ck['counters']['t'] == 2 and ck['vocab']['object_idx_to_name'][0] == '__image__' and 'hand' in ck['vocab']['object_name_to_idx']
    assert ck['vocab']['action_idx_to_name'][5].startswith('Covering')
    res = TS.main(['--checkpoint', os.path.join(out, 'latest.pt'), '--gpu_ids', '0', '--output_dir', os.path.join(str(tmp_path), 'res'),
                   '--num_clips', '1', '--frames', '4'])
    assert res['clips'] == 1 and res['edits'] >= 3 * 5
