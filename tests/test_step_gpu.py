"""End-to-end parity of the product training step on the GPU (HIP kernels through the C ABI) against
(a) the golden vectors of the unmodified reference (tests/golden/toy64_step.npz) and (b) the CPU oracle on a second
configuration.  Tolerances: losses and output tensors 1e-3 relative (BASELINE.json north_star); gradient statistics as in
tests/test_oracle_golden.py, widened to the measured conditioning of each parameter group: the same fp32 algorithm
re-run in fp64 on the CPU moves the abs-sum of the gradients by up to 3.2e-2 (flows_network), 2.1e-2 (acts_to_objs: its
gradient arrives through the layouts that feed the flow net and the warp, whose derivative is piecewise constant) and
8e-3 (rest of layout_to_video); a GPU (different summation order) cannot be closer to the reference than that, so the
bounds are 5e-2 / 5e-2 / 2e-2.  Kernel-level backward parity (2e-4) is in tests/test_ops_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(autouse=True)
def _deterministic_reductions():
    """Every comparison with the reference golden / the oracle in this file runs with ops.deterministic(): the compared value is then
    a property of the build, not of one launch's split-K atomics order (VERDICT r04 next #1b).  The default path's kernels are pinned
    one by one in tests/test_ops_gpu.py, test_halo_gpu.py, test_grad_slots_gpu.py and, as a step, by the spread report of
    tests/test_configs_gpu.py."""
    from ag2video_amd import ops
    with ops.deterministic():
        yield


def _setup():
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    gold = np.load(os.path.join(HERE, 'golden', 'toy64_step.npz'), allow_pickle=False)
    opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags'])))
    return gold, opt, json.loads(str(gold['meta.toy']))


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_train_step_matches_reference_golden(prec, device):
    """Both conv arithmetic modes (default bf16x3 split products; exact fp32 MFMA) against the reference golden vectors."""
    from ag2video_amd import ops
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    ops.set_conv_precision(prec)
    gold, opt, toy = _setup()
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=toy['wseed'])
    worst = {}
    for it in range(2):
        batch = ts.to_device(make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed'] + it))
        model_out, GL, DL = ts.gd_step(batch)
        p = 'it%d.' % it
        for k, v in list(GL.items()) + list(DL.items()):
            key = p + ('G_loss.' if k in GL else 'D_loss.') + k
            ref = float(gold[key])
            ltol = 1e-3 if it == 0 else 2e-2     # iteration 1 is chaotic (Adam sign steps): test_oracle_golden.py CHAOS note
            assert abs(v.item() - ref) <= ltol * max(1.0, abs(ref)), (key, v.item(), ref)
        if it == 0:
            for name, got in zip(('imgs_pred', 'boxes_pred', 'flows_pred'), model_out[:3]):
                ref = gold[name]
                err = np.abs(got.detach().cpu().numpy() - ref).max()
                worst[name] = err / np.abs(ref).max()
                assert err <= 1e-3 * np.abs(ref).max(), (name, err, np.abs(ref).max())
            assert np.mean(model_out[3].cpu().numpy() != gold['conf_pred']) < 2e-3
            assert (model_out[4][1].cpu().numpy() == gold['temporal_triplets']).all()
            bad = []
            for n, prm in ts.model.named_parameters():
                key = 'G_grad.' + n
                if prm.grad is None or key not in gold.files:
                    continue
                got, ref = _stats(prm.grad), gold[key]
                rel = 5e-2 if ('flows_network' in n or n.startswith('acts_to_objs')) else 2e-2
                floor = 2e-6 * prm.numel()
                if abs(got[1] - ref[1]) > rel * abs(ref[1]) + floor:
                    bad.append((n, got[1], ref[1]))
            assert not bad, bad[:10]
            g = ts.model.layout_to_video.module.netG.conv_img.weight.grad.cpu().numpy()
            ref = gold['G_grad_full.netG.conv_img.weight']
            assert np.abs(g - ref).max() <= 5e-3 * np.abs(ref).max()
            badd = []
            for n, prm in ts.discriminator.img_discriminator.named_parameters():
                key = 'D_grad.' + n
                if prm.grad is None or key not in gold.files:
                    continue
                got, ref = _stats(prm.grad), gold[key]
                if abs(got[1] - ref[1]) > 2e-2 * abs(ref[1]) + 2e-6 * prm.numel():
                    badd.append((n, got[1], ref[1]))
            assert not badd, badd[:10]
        gb = ts.to_device(make_clip_batch(opt.vocab, toy['batch'], toy['graph_frames'], tuple(toy['size']),
                                          seed=toy['seed'] + 100 + it))
        boxes_pred, GGL = ts.graph_step(gb)
        for k, v in GGL.items():
            ref = float(gold[p + 'graph_loss.' + k])
            assert abs(v.item() - ref) <= (1e-3 if it == 0 else 2e-2) * max(1.0, abs(ref)), (k, v.item(), ref)
        if it == 0:
            ref = gold['graph.boxes_pred']
            assert np.abs(boxes_pred.detach().cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
        # ALL parameters (after Adam), spectral-norm u / v and batch-norm running statistics after this iteration's optimizer steps,
        # as tests/test_oracle_golden.py holds the oracle: ~600 keys.  A wrong number of power iterations or running-stat updates
        # per step (one per forward CALL: 2 frame passes for G, 4 passes for D) shows up here as an O(1) deviation of that key.
        # Iteration 0 at 1e-3 (+ Adam's lr-per-element floor for parameters whose true gradient is zero); iteration 1 at 3e-2 (chaotic).
        adam_floor = 1.2 * opt.learning_rate * (it + 1)
        srel = 1e-3 if it == 0 else 3e-2
        bad_state, n_keys = [], 0
        for tag, sd in (('G_state.', ts.model.state_dict()), ('D_state.', ts.discriminator.img_discriminator.state_dict())):
            for k, v in sd.items():
                key = p + tag + k
                assert key in gold.files, key
                got, ref = _stats(v.float()), gold[key]
                n_keys += 1
                n = v.numel()
                for i, floor in ((1, adam_floor * n), (2, adam_floor * n ** 0.5)):
                    if abs(got[i] - ref[i]) > srel * abs(ref[i]) + floor:
                        bad_state.append((key, i, got[i], ref[i]))
        assert n_keys == 492 + 56, n_keys
        assert not bad_state, (len(bad_state), bad_state[:10])
    print('worst output errors (relative to max) [%s]:' % prec, worst)
    ops.set_conv_precision('bf16x3')


_BP_O64 = {}


def _bp_prev_o64(opt, toy):
    """The fp64 oracle's bp_prev step on the fixture's clip and weights (CPU, ~10 s, once per session): the EXACT result of the graph."""
    if 'r' not in _BP_O64:
        from ag2video_amd.data.synthetic import make_clip_batch
        from oracle.step import OracleTrainer, toy_key_shapes
        kg, kd, _ = toy_key_shapes()
        tr = OracleTrainer(opt, kg, kd, toy['wseed'], dtype=torch.float64)
        mo, GL, DL = tr.gd_step(make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed']), apply_updates=False)
        _BP_O64['r'] = (mo[0].detach().numpy(), mo[2].detach().numpy(), {n: g for n, g in tr.last_G_grads.items() if g is not None},
                        {n: g for n, g in tr.last_D_grads.items() if g is not None})
    return _BP_O64['r']


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_train_step_bp_prev_matches_reference_golden(prec, device):
    """A TRAINING step with --bp_prev 1 (reference generator.py:68-71: the generated previous frame is fed back, so frame t's loss
    back-propagates through frames t-1, t-2 -- the sequential generator path under autograd; VERDICT r05 missing #2) against
    tests/golden/bp_prev.npz, written from the imported reference by tests/golden/make_golden_bp_prev.py: 3 frames -- frame 2 is
    generated from the GENERATED frame 1 (one level of feedback).

    Gates.  Feeding a generated frame back through the flow warp amplifies rounding ~10x per level: the REFERENCE's own fp32 result
    (the golden; the fp32 oracle reproduces it bit for bit on the build host) sits 2e-4 / 4e-3 of max|imgs_pred| from the exact (fp64)
    result at generated frames 1 / 2 (relative L2 1e-4 / 1.1e-3; a third generated frame: 3.6e-2 / 8e-3 -- which is why the fixture stops
    at one level) and its gradient abs-sums up to 9 % (median 1 %) from the exact ones -- DESIGN.md 13.  So: losses 1e-3 against the
    golden; frame 1 (no feedback yet) 1e-3 against the golden; every frame against the EXACT result within 3x (relative L2) / 4x
    (max-norm) of the golden's own distance from it; gradients: see below."""
    from ag2video_amd import ops
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    gold = np.load(os.path.join(HERE, 'golden', 'bp_prev.npz'), allow_pickle=False)
    opt = make_opt(cater_vocab(), json.loads(str(gold['meta.flags'])))
    toy = json.loads(str(gold['meta.toy']))
    assert opt.bp_prev == 1 and float(gold['teacher_forced.imgs_pred_maxdiff']) > 0.1
    opt.gpu_ids = [0]
    img64, flow64, gG64, gD64 = _bp_prev_o64(opt, toy)
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        ts = TrainStep(opt, device, weights_seed=toy['wseed'])
        batch = ts.to_device(make_clip_batch(opt.vocab, toy['batch'], toy['frames'], tuple(toy['size']), seed=toy['seed']))
        model_out, GL, DL = ts.gd_step(batch, apply_updates=False)
        for k, v in list(GL.items()) + list(DL.items()):
            ref = float(gold[('G_loss.' if k in GL else 'D_loss.') + k])
            assert abs(v.item() - ref) <= 1e-3 * max(1.0, abs(ref)), (k, v.item(), ref)
        img = model_out[0].detach().double().cpu().numpy()
        gimg = gold['imgs_pred'].astype(np.float64)
        scale = np.abs(img64).max()
        report = {}
        for t in range(1, img.shape[1]):
            def errs(a, b):
                d = a[:, t] - b[:, t]
                return float(np.abs(d).max() / scale), float(np.linalg.norm(d) / np.linalg.norm(b[:, t]))
            f_max, f_l2 = errs(gimg, img64)                    # the reference arithmetic's own distance from the exact result
            e_max, e_l2 = errs(img, img64)
            g_max, _ = errs(img, gimg)
            report['frame %d' % t] = dict(vs_exact=(e_max, e_l2), golden_vs_exact=(f_max, f_l2), vs_golden_max=g_max)
            assert e_l2 <= max(1e-3, 3.0 * f_l2) and e_max <= max(1e-3, 4.0 * f_max), (t, report)
            if t == 1:
                assert g_max <= 1e-3, (t, report)
        fl = model_out[2].detach().double().cpu().numpy()
        f_fl = float(np.abs(gold['flows_pred'] - flow64).max() / np.abs(flow64).max())
        e_fl = float(np.abs(fl - flow64).max() / np.abs(flow64).max())
        report['flows'] = (e_fl, f_fl)
        assert e_fl <= max(1e-3, 4.0 * f_fl), report
        assert np.mean(model_out[3].cpu().numpy() != gold['conf_pred']) < 5e-3
        # gradient abs-sums against the exact ones.  The reference's own fp32 gradients (the golden) deviate from the exact ones by up to
        # 9 % for parameters in front of the generator and 1 % in the median (one feedback level through the warp of a white-noise frame):
        # per parameter the gate is 2.5 x the LARGEST deviation the golden shows in the parameter's group (never below the teacher-forced
        # bounds); over all parameters the MEDIAN deviation -- the low-variance statistic -- must stay within 3 x the golden's median.
        rows = []
        for n, prm in ts.model.named_parameters():
            key = 'G_grad.' + n
            if prm.grad is None or key not in gold.files or n not in gG64:
                continue
            exact = float(gG64[n].abs().sum())
            if exact <= 2e-6 * prm.numel():
                continue
            front = 'flows_network' in n or n.startswith('acts_to_objs')
            rows.append((n, front, abs(_stats(prm.grad)[1] - exact) / exact, abs(float(gold[key][1]) - exact) / exact))
        assert len(rows) > 150, len(rows)
        gfloor = {f: max(r[3] for r in rows if r[1] == f) for f in (True, False)}
        bad = [r for r in rows if r[2] > max(5e-2 if r[1] else 2e-2, 2.5 * gfloor[r[1]])]
        med_gpu, med_gold = float(np.median([r[2] for r in rows])), float(np.median([r[3] for r in rows]))
        report['gradient abs-sum deviation from exact'] = dict(worst=max(rows, key=lambda r: r[2]), median=med_gpu, golden_median=med_gold,
                                                               golden_worst_front=gfloor[True], golden_worst_decoder=gfloor[False])
        assert not bad, (bad[:10], report)
        assert med_gpu <= max(1e-2, 3.0 * med_gold), report
        drows = []
        for n, prm in ts.discriminator.img_discriminator.named_parameters():
            key = 'D_grad.' + n
            if prm.grad is None or key not in gold.files or n not in gD64:
                continue
            exact = float(gD64[n].abs().sum())
            if exact > 2e-6 * prm.numel():
                drows.append((n, abs(_stats(prm.grad)[1] - exact) / exact, abs(float(gold[key][1]) - exact) / exact))
        dfloor = max(r[2] for r in drows)
        badd = [r for r in drows if r[1] > max(2e-2, 2.5 * dfloor)]
        report['D gradient abs-sum deviation from exact'] = dict(worst=max(drows, key=lambda r: r[1]), golden_worst=dfloor)
        assert len(drows) > 30 and not badd, (badd[:10], report)
        print('bp_prev step [%s] (max-norm / max, relative L2):' % prec, report)
    finally:
        ops.set_conv_precision(old)


@pytest.mark.parametrize("cfg", [dict(side=64, frames=4, batch=3, flags=['--ngf', '16', '--ndf', '8', '--nff', '8',
                                                                       '--gconv_hidden_dim', '128'])], ids=['64px_f4_b3'])
def test_train_step_matches_oracle_other_shape(cfg, device):
    """A second shape (odd batch, 4 frames, wider nets) against the CPU oracle run in the same process."""
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    from oracle.step import OracleTrainer
    flags = ['--no_vgg_loss', '--image_size', '%d,%d' % (cfg['side'], cfg['side']), '--batch_size', str(cfg['batch'])] + cfg['flags']
    opt = make_opt(cater_vocab(), flags + ['--gpu_ids', '0'])
    ts = TrainStep(opt, device, weights_seed=3)
    shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
    shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
    orc = OracleTrainer(opt, shapes_G, shapes_D, 3)
    cpu_batch = make_clip_batch(opt.vocab, cfg['batch'], cfg['frames'], (cfg['side'], cfg['side']), seed=77, max_objs=7)
    mo_r, GL_r, DL_r = orc.gd_step(cpu_batch)
    mo, GL, DL = ts.gd_step(ts.to_device(cpu_batch))
    for k in GL_r:
        assert abs(GL[k].item() - GL_r[k].item()) <= 1e-3 * max(1.0, abs(GL_r[k].item())), (k, GL[k].item(), GL_r[k].item())
    for k in DL_r:
        assert abs(DL[k].item() - DL_r[k].item()) <= 1e-3 * max(1.0, abs(DL_r[k].item())), (k, DL[k].item(), DL_r[k].item())
    for i, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred')):
        ref = mo_r[i].detach().numpy()
        err = np.abs(mo[i].detach().cpu().numpy() - ref).max()
        assert err <= 1e-3 * np.abs(ref).max(), (name, err)


def _pair(flags, vocab, wseed, device):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.engine import TrainStep
    from oracle.step import OracleTrainer
    opt = make_opt(vocab, flags + ['--gpu_ids', '0', '--no_vgg_loss'])
    ts = TrainStep(opt, device, weights_seed=wseed)
    shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
    shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
    return opt, ts, OracleTrainer(opt, shapes_G, shapes_D, wseed)


def test_inference_path_matches_oracle(device):
    """test_mode=True (scripts/test.py's path): frames are generated autoregressively from the model's own previous output.
    Layout boxes are the ground truth: with closed-form random weights the *predicted* boxes are garbage (negative widths,
    far outside the image), which makes the comparison a test of conditioning rather than of the code.  Norm layers stay in batch-statistics mode: with freshly initialised running
    statistics (mean 0, var 1) nothing normalises the closed-form weights and both sides overflow; the eval-mode norm
    kernels are covered by test_ops_gpu.py::test_norm_eval_mode."""
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from oracle import model as M
    opt, ts, orc = _pair(['--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
                          '--gconv_hidden_dim', '64'], cater_vocab(), 9, device)
    clip = make_clip_batch(opt.vocab, 2, 4, (64, 64), seed=5)
    imgs, objs, boxes, triplets, actions = clip[:5]
    with torch.no_grad():
        ref = M.generator_forward(orc.G, opt, imgs, objs, triplets, actions, boxes, training=True, use_gt=True, test_mode=True)
        d = ts.to_device(clip)
        got = ts.model(d[0], d[1], d[3], d[4], boxes_gt=d[2], test_mode=True, use_gt=True)
    # The autoregressive loop feeds each generated frame back through the flow net and the warp; with random weights it
    # amplifies any rounding difference ~15x per frame (measured on the CPU between the dense and the low-rank evaluation
    # of the same convolution: 7e-4 / 8e-3 / 1.2e-1 absolute at frames 1 / 2 / 3).  Frame 1 (fed by the ground-truth frame)
    # is therefore held to 1e-3, frame 2 (the first to consume a generated frame) to 2e-2; a plumbing error would be O(1).
    r_boxes = ref[1].numpy()
    assert np.abs(got[1].cpu().numpy() - r_boxes).max() <= 1e-3 * np.abs(r_boxes).max()
    r_img, g_img = ref[0].numpy(), got[0].cpu().numpy()
    assert np.isfinite(r_img).all()
    scale = np.abs(r_img).max()
    assert np.abs(g_img[:, 0] - r_img[:, 0]).max() == 0.0
    assert np.abs(g_img[:, 1] - r_img[:, 1]).max() <= 1e-3 * scale
    assert np.abs(g_img[:, 2] - r_img[:, 2]).max() <= 2e-2 * scale
    r_fl, g_fl = ref[2].numpy(), got[2].cpu().numpy()
    assert np.abs(g_fl[:, 0] - r_fl[:, 0]).max() <= 1e-3 * np.abs(r_fl).max()


def test_single_attribute_vocab_step(device):
    """Something-Else shaped vocabulary: one attribute => semantic_nc = 128, no attribute_fc_gen, 2-3 objects per clip."""
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import smth_vocab
    opt, ts, orc = _pair(['--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
                          '--gconv_hidden_dim', '64'], smth_vocab(12), 4, device)
    assert opt.semantic_nc == 128
    clip = make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=21, min_objs=2, max_objs=3, max_actions=2)
    _, GL_r, DL_r = orc.gd_step(clip)
    _, GL, DL = ts.gd_step(ts.to_device(clip))
    for got, ref in ((GL, GL_r), (DL, DL_r)):
        for k in ref:
            assert abs(got[k].item() - ref[k].item()) <= 1e-3 * max(1.0, abs(ref[k].item())), (k, got[k].item(), ref[k].item())


def test_shared_layout_terms_match_separate_evaluation(device):
    """LossModel evaluates the discriminator's layout terms once per step (in the generator loss) and reuses them, with their
    graph, in the discriminator loss of the same batch.  Against separate evaluations (opt.no_shared_layout_terms) on the SAME
    generator output and module state: same losses, same discriminator gradients.  (Whole steps are not compared: the
    generator's split-K atomics perturb its output in the last bit from run to run, and at random init that flips LeakyReLU /
    hinge kinks inside D -- percent-level jumps in single gradients in BOTH modes, tools/debug_shared_terms3.py.)  Then two
    updating steps count the evaluations: one per step, i.e. the terms are dropped once consumed."""
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    flags = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '8', '--ndf', '8', '--nff', '8',
             '--gconv_hidden_dim', '64', '--gpu_ids', '0']
    opt = make_opt(cater_vocab(), flags)
    ts = TrainStep(opt, device, weights_seed=5)
    D = ts.discriminator.img_discriminator
    calls = [0]
    inner = D.layout_terms
    def counted(*a, **k):
        calls[0] += 1
        return inner(*a, **k)
    D.layout_terms = counted
    b0 = ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=11))
    imgs, objs, boxes, triplets, actions, _ = b0
    with torch.no_grad():
        mo = ts.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
    bufs = {n: b.detach().clone() for n, b in D.named_buffers()}        # spectral-norm vectors advance in every D pass
    res = []
    for separate in (False, True):
        opt.no_shared_layout_terms = separate
        with torch.no_grad():
            for n, b in D.named_buffers():
                b.copy_(bufs[n])
        calls[0] = 0
        GL = ts.gans_model(b0, mo, mode="compute_generator_loss")
        DL = ts.gans_model(b0, mo, mode="compute_discriminator_loss")
        ts.discriminator.optimizer_d_img.zero_grad()
        DL["total_img_loss"].backward()
        res.append((calls[0], {**{k: v.item() for k, v in GL.items()}, **{k: v.item() for k, v in DL.items()}},
                    {n: p.grad.detach().clone() for n, p in D.named_parameters() if p.grad is not None}))
    (c_sh, l_sh, g_sh), (c_se, l_se, g_se) = res
    assert c_sh == 1 and c_se == 2, (c_sh, c_se)
    for k in l_se:
        assert abs(l_sh[k] - l_se[k]) <= 1e-6 * max(1.0, abs(l_se[k])), (k, l_sh[k], l_se[k])
    assert set(g_sh) == set(g_se) and len(g_se) > 20
    for k in g_se:
        err = (g_sh[k] - g_se[k]).abs().max().item()
        assert err <= 1e-4 * g_se[k].abs().max().item() + 1e-9, (k, err, g_se[k].abs().max().item())
    opt.no_shared_layout_terms = False
    calls[0] = 0
    for step in range(2):
        ts.gd_step(ts.to_device(make_clip_batch(opt.vocab, 2, 3, (64, 64), seed=12 + step)))
    assert calls[0] == 2 and ts.gans_model.module._terms_cache is None, calls[0]
