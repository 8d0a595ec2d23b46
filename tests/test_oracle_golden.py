"""Pins the CPU oracle (oracle/model.py) against golden vectors produced by the unmodified reference modules
(tests/golden/toy64_step.npz; generator script tests/golden/make_golden.py): two full training iterations
(G step, D step, graph step) at the toy configuration 64x64, B=2, F=3.

Tolerances: losses 2e-4 relative; output tensors max-abs error <= 2e-4 * max|ref|; gradient / post-step parameter
statistics (abs-sum, L2) 2e-3 relative plus an absolute floor of 2e-6 per element (conv biases in front of a batch norm
have an exactly-zero true gradient: what is stored there is rounding noise).  Gradients that pass through the optical
-flow warp (every flows_network parameter) are held to 5e-2 only: d(warp)/d(flow) is piecewise constant in the sampling
cell, and the same algorithm run in fp32 and in fp64 already differs by 1-3 % on them (measured: down_flow.0.1.bias
abs-sum 2.749 fp32 / 2.840 fp64 / 2.759 reference), so a tighter bound would test the rounding, not the restatement.  Full gradient tensors are compared at 5e-3 of their
max: the L1 / hinge / LeakyReLU derivatives are sign functions, so ulp-level forward differences flip individual terms.

CHAOS note (iteration 1): Adam's first update is lr*sign(g) for every element, so every weight whose gradient is below
the rounding noise moves by +-1e-4 in a direction that depends on summation order.  Measured: the very same oracle run
on another host CPU (the GPU box: different core count / oneDNN blocking) gives it1 GAN_Feat 4.8032 against 4.7847 in
the fixture (0.39 %), and the GPU differs from *that* oracle by 0.04 %.  Iteration 1 is therefore only sanity-checked at
2e-2; parity is gated on iteration 0 (same weights bit for bit), where everything is held to the bounds above.
"""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'toy64_step.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD, allow_pickle=False)


def _opt(gold):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    flags = json.loads(str(gold['meta.flags']))
    return make_opt(cater_vocab(), flags), json.loads(str(gold['meta.toy']))


def _stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def _close_stats(name, got, ref, rel=2e-3, numel=1, per_elem=2e-6):
    if 'flows_network' in name and 'grad' in name:
        rel = 5e-2
    floors = (0.0, per_elem * numel, per_elem * numel ** 0.5)
    for i in (1, 2):
        assert abs(got[i] - ref[i]) <= rel * abs(ref[i]) + floors[i], "%s stat %d: %r vs %r" % (name, i, got, ref)


def _close(name, got, ref, tol=2e-4):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, name
    err = np.abs(got - ref).max()
    assert err <= tol * (np.abs(ref).max() + 1e-12), "%s: err %.3e scale %.3e" % (name, err, np.abs(ref).max())


def test_state_key_tables():
    from oracle.step import toy_key_shapes
    g, d, lm = toy_key_shapes()
    assert len(g) == 492 and len(d) == 56 and len(lm) == 112      # SURVEY.md 8b [probe]


def test_oracle_matches_reference_two_iterations(gold):
    torch.set_num_threads(8)
    from ag2video_amd.data.synthetic import make_clip_batch
    from oracle.step import OracleTrainer, toy_key_shapes
    opt, toy = _opt(gold)
    kg, kd, _ = toy_key_shapes()
    tr = OracleTrainer(opt, kg, kd, toy['wseed'])
    for it in range(2):
        batch = make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed'] + it)
        model_out, GL, DL = tr.gd_step(batch)
        p = 'it%d.' % it
        ltol = 2e-4 if it == 0 else 2e-2   # iteration 1: see CHAOS note in the module docstring
        for k, v in GL.items():
            ref = float(gold[p + 'G_loss.' + k])
            assert abs(v.item() - ref) <= ltol * max(1.0, abs(ref)), (p, k, v.item(), ref)
        for k, v in DL.items():
            ref = float(gold[p + 'D_loss.' + k])
            assert abs(v.item() - ref) <= ltol * max(1.0, abs(ref)), (p, k, v.item(), ref)
        if it == 0:
            imgs_pred, boxes_pred, flows, conf, adata = model_out
            _close('imgs_pred', imgs_pred.detach().numpy(), gold['imgs_pred'])
            _close('boxes_pred', boxes_pred.detach().numpy(), gold['boxes_pred'])
            _close('flows_pred', flows.detach().numpy(), gold['flows_pred'], 5e-4)
            assert np.mean(conf.numpy() != gold['conf_pred']) < 1e-3
            assert (adata[1].numpy() == gold['temporal_triplets']).all()
            _close('relative_timesteps', adata[2].numpy(), gold['relative_timesteps'], 1e-6)
            for n, g in tr.last_G_grads.items():
                key = 'G_grad.' + n
                if g is None:
                    assert key not in gold.files, n
                    continue
                if key in gold.files:
                    _close_stats(key, _stats(g), gold[key], 2e-3, g.numel())
            L2V = 'layout_to_video.module.'
            _close('grad conv_img', tr.last_G_grads[L2V + 'netG.conv_img.weight'].numpy(),
                   gold['G_grad_full.netG.conv_img.weight'], 5e-3)
            _close('grad up_3.conv_1', tr.last_G_grads[L2V + 'netG.up_3.conv_1.weight_orig'].numpy(),
                   gold['G_grad_full.up_3.conv_1.weight_orig'], 5e-3)
            _close('grad acts_embeddings', tr.last_G_grads['acts_to_objs.module.acts_embeddings.weight'].numpy(),
                   gold['G_grad_full.acts_to_objs.acts_embeddings'], 5e-3)
            _close('grad conv_flow', tr.last_G_grads[L2V + 'flows_network.conv_flow.0.weight'].numpy(),
                   gold['G_grad_full.flows.conv_flow.weight'], 5e-2)
            for n, g in tr.last_D_grads.items():
                key = 'D_grad.' + n
                if g is not None and key in gold.files:
                    _close_stats(key, _stats(g), gold[key], 2e-3, g.numel())
            _close('grad D model4', tr.last_D_grads['discriminator_1.model4.0.weight'].numpy(),
                   gold['D_grad_full.discriminator_1.model4.0.weight'], 5e-3)
        gb = make_clip_batch(opt.vocab, toy['batch'], toy['graph_frames'], tuple(toy['size']), seed=toy['seed'] + 100 + it)
        boxes_pred_g, GGL = tr.graph_step(gb)
        for k, v in GGL.items():
            ref = float(gold[p + 'graph_loss.' + k])
            assert abs(v.item() - ref) <= ltol * max(1.0, abs(ref)), (p, k, v.item(), ref)
        if it == 0:
            _close('graph.boxes_pred', boxes_pred_g.detach().numpy(), gold['graph.boxes_pred'])
        # parameters, spectral-norm u/v and BN running statistics after the optimizer steps.  Adam's first updates are
        # +-lr per element whatever the gradient's size, so a parameter whose true gradient is zero (a conv bias in front
        # of a batch norm: its stored gradient is rounding noise) moves by lr in a noise-determined direction: the
        # per-element floor is lr per step taken.
        adam_floor = 1.2 * opt.learning_rate * (it + 1)
        srel = 1e-3 if it == 0 else 2e-2      # iteration 1: see CHAOS note in the module docstring
        for n, t in tr.G.items():
            _close_stats(p + 'G_state.' + n, _stats(t.float()), gold[p + 'G_state.' + n], srel, t.numel(), adam_floor)
        for n, t in tr.D.items():
            _close_stats(p + 'D_state.' + n, _stats(t.float()), gold[p + 'D_state.' + n], srel, t.numel(), adam_floor)


def test_oracle_fp64_mode_and_committed_noise_floor():
    """oracle/step.py dtype=torch.float64 (the exact-arithmetic run of the SAME fp32 weights and clip that the GPU step tests compare with
    beside the fp32 oracle) on the toy configuration: every tensor of the iteration is double, the fp32 oracle sits within fp32 rounding of
    it, and tests/golden/noise_floor.json (tests/golden/make_noise_floor.py) holds a floor for every step case the GPU tests gate."""
    import json
    import sys
    sys.path.insert(0, HERE)
    import _cases as C
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from oracle.step import OracleTrainer, toy_key_shapes
    gold = np.load(os.path.join(HERE, 'golden', 'toy64_step.npz'), allow_pickle=False)
    opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags'])))
    toy = json.loads(str(gold['meta.toy']))
    sG, sD, _ = toy_key_shapes()
    clip = make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed'])
    res = {}
    for dt in (torch.float32, torch.float64):
        tr = OracleTrainer(opt, sG, sD, toy['wseed'], dtype=dt)
        mo, GL, DL = tr.gd_step(clip)
        assert all(t.dtype == dt for t in mo[:4]) and all(v.dtype == dt for v in GL.values())
        assert all(t.dtype == dt for t in tr.G.values() if t.is_floating_point())
        res[dt] = (mo, GL, DL)
    assert torch.get_default_dtype() == torch.float32                   # the fp64 run restores the default dtype
    for i in range(3):
        mx, l2 = C.out_errors(res[torch.float32][0][i], res[torch.float64][0][i])
        assert mx <= 1e-3 and l2 <= 3e-4, (i, mx, l2)
    for k, v in res[torch.float64][1].items():
        assert abs(float(res[torch.float32][1][k]) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
    floor = json.load(open(os.path.join(HERE, 'golden', 'noise_floor.json')))
    for case in C.STEP_CASES:
        smp = floor[case]['imgs_pred_o32_samples_vs_o64']
        # (c1bp: the generated frame is fed back through the warp twice -- the fp32 oracle itself sits 1e-2 / 5e-2 from fp64 there)
        cap_l2, cap_max = (5e-2, 0.2) if case.endswith('bp') else (1e-3, 5e-3)
        assert len(smp) >= 3 and all(0 < v['l2'] < cap_l2 and 0 < v['max'] < cap_max for v in smp.values()), (case, smp)
        assert set(floor[case]['o32_vs_o64']['grad']) >= {'front', 'decoder', 'D'}
    assert floor['vgg_grad']['reference_golden_vs_o64']['l2'] < 1e-5


def test_oracle_matches_reference_bp_prev_training_step():
    """--bp_prev 1 (reference generator.py:68-71: training feeds the GENERATED previous frame back, so frame t's loss back-propagates
    through frames t-1, t-2): the oracle against tests/golden/bp_prev.npz, written by tests/golden/make_golden_bp_prev.py from the
    imported reference -- 3 frames: frame 2 is generated from the generated frame 1.  Tolerances as in the teacher-forced test above; the fixture also records
    that the mode matters (imgs_pred differs from the teacher-forced pass by O(1))."""
    torch.set_num_threads(8)
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from oracle.step import OracleTrainer, toy_key_shapes
    gold = np.load(os.path.join(HERE, 'golden', 'bp_prev.npz'), allow_pickle=False)
    assert float(gold['teacher_forced.imgs_pred_maxdiff']) > 0.1
    opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags'])))
    assert opt.bp_prev == 1
    toy = json.loads(str(gold['meta.toy']))
    kg, kd, _ = toy_key_shapes()
    tr = OracleTrainer(opt, kg, kd, toy['wseed'])
    batch = make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed'])
    model_out, GL, DL = tr.gd_step(batch, apply_updates=False)
    for k, v in GL.items():
        ref = float(gold['G_loss.' + k])
        assert abs(v.item() - ref) <= 2e-4 * max(1.0, abs(ref)), (k, v.item(), ref)
    for k, v in DL.items():
        ref = float(gold['D_loss.' + k])
        assert abs(v.item() - ref) <= 2e-4 * max(1.0, abs(ref)), (k, v.item(), ref)
    _close('imgs_pred', model_out[0].detach().numpy(), gold['imgs_pred'])
    _close('flows_pred', model_out[2].detach().numpy(), gold['flows_pred'], 5e-4)
    assert np.mean(model_out[3].numpy() != gold['conf_pred']) < 1e-3
    n_checked = 0
    for n, g in tr.last_G_grads.items():
        key = 'G_grad.' + n
        if g is None:
            assert key not in gold.files, n
        elif key in gold.files:
            _close_stats(key, _stats(g), gold[key], 2e-3, g.numel())
            n_checked += 1
    assert n_checked > 150
    L2V = 'layout_to_video.module.'
    _close('grad conv_img', tr.last_G_grads[L2V + 'netG.conv_img.weight'].numpy(), gold['G_grad_full.netG.conv_img.weight'], 5e-3)
    _close('grad up_3.conv_1', tr.last_G_grads[L2V + 'netG.up_3.conv_1.weight_orig'].numpy(), gold['G_grad_full.up_3.conv_1.weight_orig'], 5e-3)
    _close('grad fc', tr.last_G_grads[L2V + 'netG.fc.weight'].numpy(), gold['G_grad_full.netG.fc.weight'], 5e-3)
    _close('grad conv_flow', tr.last_G_grads[L2V + 'flows_network.conv_flow.0.weight'].numpy(), gold['G_grad_full.flows.conv_flow.weight'], 5e-2)
    for n, g in tr.last_D_grads.items():
        key = 'D_grad.' + n
        if g is not None and key in gold.files:
            _close_stats(key, _stats(g), gold[key], 2e-3, g.numel())


@pytest.mark.parametrize("fixture", ['c2', 'c3h', 'c5h'])
def test_oracle_c2_full_size_fixture_is_consistent(fixture):
    """tests/golden/oracle_c2.npz (tests/golden/make_oracle_c2.py: BASELINE.json's configs[1] at full size through the oracle, fp32 and
    fp64): complete -- every loss, the 492 + 56 state keys, gradient statistics of both nets, for both oracles -- and self-consistent:
    the fp32 oracle sits within fp32-level distance of the fp64 one (the floors the GPU test derives its gates from)."""
    import sys
    sys.path.insert(0, HERE)
    import _cases as C
    gold = np.load(os.path.join(HERE, 'golden', 'oracle_%s.npz' % fixture), allow_pickle=False)
    meta = json.loads(str(gold['meta']))
    side, frames, batch = C.FULL_CASES[fixture]
    assert meta['side_frames_batch'] == [side, frames, batch] and meta['case'] == fixture
    assert (meta['weights_seed'], meta['clip_seed'], meta['graph_clip_seed']) == (C.WEIGHTS_SEED, C.CLIP_SEED, C.GRAPH_CLIP_SEED)
    for tag in ('o32', 'o64'):
        nG, nD = sum(k.startswith(tag + '.state.G.') for k in gold.files), sum(k.startswith(tag + '.state.D.') for k in gold.files)
        # (CATER: 4 attributes => the reference's 492 + 56 keys; Something-Else: one attribute, AttributeEmbeddings has no mixing Linear)
        assert (nG, nD) == (492, 56) if not fixture.startswith('c5') else (nG >= 470 and nD >= 50), (nG, nD)
        assert sum(k.startswith(tag + '.G_grads.') for k in gold.files) > 240 and sum(k.startswith(tag + '.D_grads.') for k in gold.files) > 35
        assert {k.split('.', 2)[2] for k in gold.files if k.startswith(tag + '.GL.')} >= {'GAN_Img', 'GAN_Feat', 'loss_F_Warp', 'total_loss'}
        assert gold[tag + '.imgs_sub'].shape == (batch, frames, 3, side // 4, side // 4) and gold[tag + '.flows_sub'].shape == (batch, frames, 2, side // 4, side // 4)
    for k in gold.files:
        if k.startswith('o64.GL.') or k.startswith('o64.DL.') or k.startswith('o64.GG.'):
            a, b = float(gold['o32' + k[3:]]), float(gold[k])
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
    d = gold['o32.imgs_sub'] - gold['o64.imgs_sub']
    fmax, fl2 = np.abs(d).max() / float(gold['o64.imgs_absmax']), np.linalg.norm(d) / np.linalg.norm(gold['o64.imgs_sub'])
    assert 0 < fl2 < 1e-3 and 0 < fmax < 5e-3, (fmax, fl2)
    assert np.abs(gold['o32.boxes_pred'] - gold['o64.boxes_pred']).max() < 1e-5
