"""BASELINE.json's configurations through the HIP path (VERDICT r01 "configs never run through the HIP path": c1, c2, c5).

* c1 (64x64, 4 frames, batch 1, DEFAULT widths: ngf 64 / ndf 64 / nff 32 / hidden 512 -> the 1024-channel layers, spectral norm on
  1024 x 9216 matrices, K = 9*1024 reductions): one full iteration (G step, D step, graph step) on the GPU against the CPU
  oracle on the same seeded clip and closed-form weights -- losses and output tensors within 1e-3 relative, gradient statistics
  at the conditioning bounds of tests/test_step_gpu.py, every post-step state tensor (parameters after Adam, spectral-norm
  u / v, running statistics).  Both arithmetic modes.
* c2 (128x128, 8 frames, batch 4) and a c5-shaped clip (Something-Else vocabulary: one attribute => 128 layout channels, 2-3
  objects, 224x224 = 7 * 32: the non-power-of-two pyramid 7, 14, ..., 224, 16 frames): the oracle cannot run them in seconds,
  so they go through the size-independent property set of tests/test_fullsize_gpu.py at THEIR layer sizes (oracle on crops,
  adjoint identities) plus one finite training step with the documented keys.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

import _cases as C

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, device='cpu', scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


# ---- GPU step against the CPU oracle at default widths (cases: tests/_cases.py).
#
# What is compared with what.  The oracle runs twice per case: in fp32 (o32: the restatement of the reference,
# pinned to it bit-for-bit by tests/test_oracle_golden.py) and in fp64 on the SAME fp32 weights and clip (o64).  o64 is the exact result
# of the graph to ~1e-13 and does not depend on the host (thread count, MKL-DNN blocking); o32 does: two thread counts of the same fp32
# oracle differ by 2-8e-4 of max|imgs_pred| at c1 / c2p / c3p (tests/golden/noise_floor.json, `threads_1_vs_n`).  That file -- written by
# tests/golden/make_noise_floor.py, CPU only -- holds the distance of o32 from o64 for every statistic bounded here, i.e. how far the
# REFERENCE ARITHMETIC itself sits from the exact result in that norm:
#     imgs_pred   max-norm 4.1e-4 (c1) 4.2e-4 (c2p) 8.9e-4 (c3p) 1.14e-3 (c5p) 9.9e-4 (c5pT);  relative L2 0.7-1.2e-4 everywhere
#     flows_pred  max-norm 3-6e-6; boxes 1-3e-7; losses <= 1.1e-6; post-step buffers <= 1.7e-5
#     gradient abs-sum, worst parameter per group: front 0.2-2.2e-2, decoder 2-4e-3, D / T 0.6-5e-3
# The max-norm of imgs_pred is a maximum over 10^5-10^6 pixels of white-noise frames warped by a flow that carries an fp32-rounding-level
# error: fp32 itself does not hold 1e-3 in that norm at 224^2 / 256^2, so no fp32-class implementation can be asked to.
# And one evaluation's max-norm is only defined to within a factor: the fp32 oracle's OWN samples (1 / 2 / 4 / 8 threads = four summation
# orders of the same arithmetic, `imgs_pred_o32_samples_vs_o64`) span 2.5-4.3e-4 at c1, 0.89-1.16e-3 at c3p, with the SAME relative L2
# (+-3 %) -- a bound of 1.5 x one sample would fail the reference against itself (c1: 4.3e-4 > 1.5 x 2.5e-4).
# Gates on imgs_pred (floor_max / floor_l2 = the LARGEST of the oracle's samples for the case, committed in noise_floor.json):
#     relative L2 against o64     <= 2 x floor_l2(case)       (1.4-2.4e-4: the tight, low-variance gate -- HIP's error energy within 2x
#                                                              of the fp32 reference's own; measured 1.0-1.5x, the 1.5 on the exact-
#                                                              fp32 deterministic path whose 9216-term chains are summed sequentially)
#     relative L2 against o32     <= 1e-3                     (north_star's bound against the reference arithmetic itself)
#     max-norm against o64        <= max(1e-3, 1.3 x the value MEASURED for (case, arithmetic) in deterministic mode), capped by
#     max-norm against o32           round 5's max(1e-3, 2 x / 3 x floor_max(case)); the measured values are committed in
#                                    tests/golden/measured_max_norm.json (c3p: 1.18e-3 / 0.99e-3 bf16x3, 1.65e-3 / 1.22e-3 exact fp32) --
#                                    a CHANGED ACCEPTANCE CRITERION against north_star's 1e-3, stated in DESIGN.md 12.1 / README: fp32
#                                    itself does not hold 1e-3 in this norm at 224^2 / 256^2 (floors above)
# boxes / flows / losses / buffers: 1e-3 against both, as before.  Gradient statistics: the bounds of tests/test_step_gpu.py (front 5e-2,
# decoder / D / T 2e-2), >= 2.2 x the measured floors above, against both oracles.
# The step runs under ops.deterministic(): a result is then a property of the build, not of the launch (split-K atomics order).  The
# default, non-deterministic path is run once more per case as a SPREAD REPORT (printed) and gated only on the relative L2 gates.
_ORACLE_CACHE = {}
FLOOR = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'noise_floor.json')))


def _img_floor(case):
    smp = FLOOR[case]['imgs_pred_o32_samples_vs_o64'].values()
    return max(v['max'] for v in smp), max(v['l2'] for v in smp)


# Measured max-norm of imgs_pred in DETERMINISTIC mode per (case, arithmetic, oracle): tests/golden/measured_max_norm.json, written by
# tests/golden/update_measured.py from the 'MEASURED' lines this file prints on the GPU box.  The deterministic mode is reproducible
# across boxes (bit-identical on two leases), so the measured value is a property of the build and the gate is
# measured + 30 % (3 x floor left room for a regression that triples the worst pixel) -- never below north_star's
# 1e-3, never above the floor-based bound.  A case without an entry yet (new case, first run) takes the floor-based bound.
_MEASURED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'measured_max_norm.json')
MEASURED = json.load(open(_MEASURED_PATH)) if os.path.exists(_MEASURED_PATH) else {}


def _floor_scaled(case, default, *path, k=3.0):
    """max(default, k x the fp32 oracle's own distance from o64 for this statistic): a gate can only be as tight as the reference
    arithmetic itself holds the statistic.  For every teacher-forced case the floors sit far below the defaults (the gate IS the
    default); for c1bp -- the generated frame fed back through the warp of the ill-conditioned random-init decoder, twice -- the fp32
    oracle itself sits 1e-2 (relative L2) / 5e-2 (max-norm) from fp64 in imgs_pred and 15-23 % in the worst gradient abs-sum
    (tests/golden/noise_floor.json 'c1bp'), so that case's gates are the floors x k; its tight pin is the toy-width reference golden
    (tests/test_step_gpu.py::test_train_step_bp_prev_matches_reference_golden, 1e-3)."""
    v = FLOOR[case]['o32_vs_o64']
    for key in path:
        v = v.get(key, {}) if isinstance(v, dict) else {}
    return max(default, k * float(v)) if isinstance(v, (int, float)) else default


def _grad_bound(case, grp, default):
    return _floor_scaled(case, default, 'grad', grp, 'abssum_dev', k=2.5)


def _img_bounds(case, prec=None):
    """(max-norm bound against o64, against o32, relative-L2 bound against o64)."""
    fmax, fl2 = _img_floor(case)
    b64, b32 = max(1e-3, 2.0 * fmax), max(1e-3, 3.0 * fmax)
    m = MEASURED.get(case, {}).get(prec)
    if m:
        b64, b32 = min(b64, max(1e-3, 1.3 * m['o64'])), min(b32, max(1e-3, 1.3 * m['o32']))
    return b64, b32, 2.0 * fl2


def _oracle_steps(case, shapes_G, shapes_D, shapes_T=None):
    """o32 and o64 iterations of a case, computed once per session."""
    if case not in _ORACLE_CACHE:
        torch.set_num_threads(min(16, torch.get_num_threads()))
        t0 = time.time()
        o32 = C.oracle_step(case, shapes_G, shapes_D, shapes_T, torch.float32)
        t1 = time.time()
        o64 = C.oracle_step(case, shapes_G, shapes_D, shapes_T, torch.float64)
        print('%s oracle: fp32 %.0f s, fp64 %.0f s on %d threads' % (case, t1 - t0, time.time() - t1, torch.get_num_threads()))
        _ORACLE_CACHE[case] = {'o32': o32, 'o64': o64}
    return _ORACLE_CACHE[case]


def _oracle_step(case, shapes_G, shapes_D, shapes_T=None):
    return _oracle_steps(case, shapes_G, shapes_D, shapes_T)['o32']


def _check_outputs(case, prec, mo, refs, label, gate_max=True):
    """imgs / boxes / flows / conf of one step against o32 and o64; returns the measured figures."""
    b64, b32, bl2 = _img_bounds(case, prec if gate_max else None)
    got = {}
    for tag, bound_img in (('o64', b64), ('o32', b32)):
        ref = refs[tag]
        for i, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred')):
            mx, l2 = C.out_errors(mo[i], ref['out'][i])
            got[(tag, name)] = (mx, l2)
            l2_bound = bl2 if (tag == 'o64' and name == 'imgs_pred') else _floor_scaled(case, 1e-3, 'out', name, 'l2')
            assert l2 <= l2_bound, (case, prec, label, tag, name, 'relative L2', l2, 'bound', l2_bound)
            if gate_max:
                bound = bound_img if name == 'imgs_pred' else _floor_scaled(case, 1e-3, 'out', name, 'max')
                assert mx <= bound, (case, prec, label, tag, name, 'max-norm', mx, 'bound', bound)
        assert np.mean(mo[3].cpu().numpy() != ref['out'][3].float().numpy()) < _floor_scaled(case, 2e-3, 'out', 'conf_mismatch', k=10.0)
    if gate_max:
        print('MEASURED %s %s %.4e %.4e' % (case, prec, got[('o64', 'imgs_pred')][0], got[('o32', 'imgs_pred')][0]))
    print('%s [%s] %s: (max-norm / max|ref|, relative L2) %s; imgs_pred bounds: max-norm o64 %.2e o32 %.2e, L2 o64 %.2e (oracle floors %.2e / %.2e)'
          % ((case, prec, label, {'%s.%s' % k: ('%.2e' % v[0], '%.2e' % v[1]) for k, v in got.items()}, b64, b32, bl2) + _img_floor(case)))
    return got


def _grad_report(named_params, ref_stats, rel_of, allow_missing=(), per_elem=2e-6):
    """Worst deviation of sum|g| per group against one oracle's statistics; returns (bad list, worst per group)."""
    bad, worst = [], {}
    for n, prm in named_params:
        g_ref = ref_stats.get(n)
        if g_ref is None:
            assert prm.grad is None or float(prm.grad.abs().sum()) == 0.0 or n.startswith(allow_missing), n
            continue
        if prm.grad is None:
            bad.append((n, None, g_ref[1]))
            continue
        got = C.stats(prm.grad)
        grp, rel = rel_of(n)
        dev = abs(got[1] - g_ref[1]) / max(abs(g_ref[1]), 1e-30)
        if abs(g_ref[1]) > per_elem * prm.numel() and dev > worst.get(grp, (0.0, ''))[0]:
            worst[grp] = (dev, n)
        if abs(got[1] - g_ref[1]) > rel * abs(g_ref[1]) + per_elem * prm.numel():
            bad.append((n, got[1], g_ref[1]))
    return bad, worst


def _step_vs_oracle(case, prec, device):
    from ag2video_amd import ops
    from ag2video_amd.engine import TrainStep
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        opt, side, frames, batch = C.case_opt(case)
        assert (opt.ngf, opt.ndf, opt.nff, opt.gconv_hidden_dim) == (64, 64, 32, 512)
        with ops.deterministic():
            ts = TrainStep(opt, device, weights_seed=C.WEIGHTS_SEED)
        shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
        shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
        dtmp = getattr(ts.discriminator, 'temporal_discriminator', None)
        shapes_T = {k: list(v.shape) for k, v in dtmp.state_dict().items()} if dtmp is not None else None
        assert (dtmp is not None) == case.endswith('T')
        if case.startswith('c5'):
            assert opt.semantic_nc == 128 and len(opt.vocab['attributes']) == 1        # one attribute => 128 layout channels
        refs = _oracle_steps(case, shapes_G, shapes_D, shapes_T)
        clip = C.case_clip(case, opt, batch, frames, side, C.CLIP_SEED)
        with ops.deterministic():
            mo, GL, DL = ts.gd_step(ts.to_device(clip))
        if dtmp is not None:
            assert set(GL) >= {'GAN_Tmp', 'GAN_Tmp_Feat'} and set(DL) >= {'D_tmp_fake', 'D_tmp_real'}
        for tag in ('o64', 'o32'):
            for got, rf in ((GL, refs[tag]['GL']), (DL, refs[tag]['DL'])):
                assert set(got) == set(rf)
                for k in rf:
                    assert abs(got[k].item() - rf[k]) <= 1e-3 * max(1.0, abs(rf[k])), (tag, k, got[k].item(), rf[k])
        got = _check_outputs(case, prec, mo, refs, 'deterministic')
        worst = {name: got[('o32', name)][0] for name in ('imgs_pred', 'boxes_pred', 'flows_pred')}
        # gradients: abs-sum of every parameter's gradient, against both oracles (bounds: header above and tests/test_step_gpu.py;
        # correctness of the gradients themselves is pinned by tests/test_gradcheck_gpu.py (central differences) and the kernel tests)
        # (c1bp: the image gradient of flow_warp is a scatter-add even in deterministic mode, and the feedback amplifies: the rounding noise
        # of an analytically-zero bias gradient measures 2.0e-6 per element there)
        pe = 5e-6 if case.endswith('bp') else 2e-6
        for tag in ('o64', 'o32'):
            ref = refs[tag]
            bad, worst_g = _grad_report(ts.model.named_parameters(), ref['G_grads'],
                                        lambda n: ('front', _grad_bound(case, 'front', 5e-2)) if C.is_front(n) else ('decoder', _grad_bound(case, 'decoder', 2e-2)),
                                        allow_missing='acts_to_boxes', per_elem=pe)
            badd, worst_d = _grad_report(ts.discriminator.img_discriminator.named_parameters(), ref['D_grads'], lambda n: ('D', _grad_bound(case, 'D', 2e-2)))
            worst_g.update(worst_d)
            if dtmp is not None:
                badt, worst_t = _grad_report(dtmp.named_parameters(), ref['T_grads'], lambda n: ('T', 2e-2))
                worst_g.update(worst_t)
                badd += badt
            print('%s [%s] worst gradient abs-sum deviations against %s:' % (case, prec, tag), worst_g)
            assert not bad, (tag, len(bad), bad[:10])
            assert not badd, (tag, len(badd), badd[:10])
        # three full gradient TENSORS element-wise against o64 (VERDICT r05 next #6c asked for 5e-3 of max).  Measured (round 6, every case,
        # both arithmetics): netG.up_3.conv_1 0.8-4.7e-3 of max, conv_dim_in 4.0-9.7e-3 (relative L2 0.7-8e-3; the fp32 ORACLE's own
        # distance from o64 on them: up to 2.8e-3) -- and the discriminator's 512 x 256 x 4 x 4 layer 7e-3-3e-2 of max in a handful of
        # output channels at relative L2 2-4e-3.  The last one is NOT D's kernels: with the oracle's generated frames as D's input the same
        # tensor sits 1.1e-3 of max from o64 at c2p, exactly where the fp32 oracle itself sits, and 1e-6 in exact-fp32 mode (second half
        # of this block, and tests/debug/debug_full_grad.py `donly`) -- the discriminator loss's gradient is that sensitive to WHERE the
        # generated frames land (hinge / LeakyReLU kinks), and the GPU's frames differ from o64's by 4e-4 of max like the fp32 oracle's
        # own do, at other pixels.  Gates: the generator's two tensors relative L2 <= max(1e-2, 2.5 x the fp32 oracle's own) and max-norm
        # <= max(1.5e-2, 2.5 x the oracle's own); the discriminator's on the GPU's frames relative L2 <= 5e-2 (reported), on the oracle's
        # frames within 2.5 x the fp32 oracle's own distance (never below 2e-4).
        gmods = {'G_grads': dict(ts.model.named_parameters()), 'D_grads': dict(ts.discriminator.img_discriminator.named_parameters())}
        for (gtag, n), gref in refs['o64']['full_grads'].items():
            g = gmods[gtag][n].grad.detach().double().cpu()
            gref = gref.double()
            g32 = refs['o32']['full_grads'][(gtag, n)].double()
            emax = float((g - gref).abs().max() / gref.abs().max())
            el2 = float((g - gref).pow(2).sum().sqrt() / gref.pow(2).sum().sqrt())
            fmax = float((g32 - gref).abs().max() / gref.abs().max())          # the fp32 ORACLE's own distance from o64 on this tensor
            fl2 = float((g32 - gref).pow(2).sum().sqrt() / gref.pow(2).sum().sqrt())
            print('%s [%s] full gradient %s against o64: max-norm %.2e of max, relative L2 %.2e (the fp32 oracle itself: %.2e, %.2e)'
                  % (case, prec, n, emax, el2, fmax, fl2))
            if case.endswith('bp'):          # (the fp32 oracle's own gradients sit up to 0.5 from o64 in relative L2 there: reported only)
                continue
            if gtag == 'G_grads':
                assert el2 <= max(1e-2, 2.5 * fl2), (case, prec, n, 'relative L2', el2, fl2)
                assert emax <= max(1.5e-2, 2.5 * fmax), (case, prec, n, 'max-norm', emax, fmax)
            else:           # D's tensor on the GPU's own frames: kink-sensitive (2.3e-2 in relative L2 at c5p in BOTH arithmetics) -- its
                assert el2 <= 5e-2, (case, prec, n, 'relative L2', el2, fl2)      # tight pin is the oracle-frames leg below
        # ... and the discriminator's backward pinned on the ORACLE's frames: a fresh trainer (same weights), the generator loss for its two
        # D passes (the power iterations then stand where gd_step's discriminator loss finds them), the discriminator loss on o64's
        # imgs_pred, its backward -- every D parameter's gradient abs-sum at 2e-3 and the full 512 x 256 x 4 x 4 tensor element-wise
        if not case.endswith('bp') and not case.endswith('T'):
            with ops.deterministic():
                ts3 = TrainStep(opt, device, weights_seed=C.WEIGHTS_SEED)
                b3 = ts3.to_device(clip)
                mo3 = ts3.model(b3[0], b3[1], b3[3], b3[4], boxes_gt=b3[2], test_mode=False, use_gt=True)
                ts3.gans_model(b3, mo3, mode="compute_generator_loss")
                mo3 = (refs['o64']['out'][0].float().to(device),) + tuple(mo3[1:])
                DL3 = {k: v.mean() for k, v in ts3.gans_model(b3, mo3, mode="compute_discriminator_loss").items()}
                ts3.discriminator.optimizer_d_img.zero_grad()
                with ops.accumulate_into_param_grads():
                    DL3["total_img_loss"].backward()
                ops.join_direct_grad_streams()
            for k, v in refs['o64']['DL'].items():
                assert abs(DL3[k].item() - v) <= 1e-5 * max(1.0, abs(v)), ('D loss on the oracle frames', k, DL3[k].item(), v)
            bad3, worst3 = _grad_report(ts3.discriminator.img_discriminator.named_parameters(), refs['o64']['D_grads'], lambda n: ('D', 2e-3))
            print('%s [%s] D backward on the oracle frames, worst gradient abs-sum deviation against o64:' % (case, prec), worst3)
            assert not bad3, (len(bad3), bad3[:10])
            for (gtag, n), gref in refs['o64']['full_grads'].items():
                if gtag != 'D_grads':
                    continue
                g = dict(ts3.discriminator.img_discriminator.named_parameters())[n].grad.detach().double().cpu()
                gref = gref.double()
                g32 = refs['o32']['full_grads'][(gtag, n)].double()
                emax, el2 = float((g - gref).abs().max() / gref.abs().max()), float((g - gref).norm() / gref.norm())
                fmax, fl2 = float((g32 - gref).abs().max() / gref.abs().max()), float((g32 - gref).norm() / gref.norm())
                print('%s [%s] ... full gradient %s: max-norm %.2e of max, relative L2 %.2e (the fp32 oracle itself: %.2e, %.2e)'
                      % (case, prec, n, emax, el2, fmax, fl2))
                # exact-fp32 mode measures 1-4e-6 of max at c1 / c2p; where a hinge / LeakyReLU kink flips under fp32-level rounding the
                # fp32 oracle itself sits 1e-3 (c2p) ... 1e-1 (c1) of max from o64 in single elements, and so does the bf16x3 mode
                assert emax <= max(2e-4, 2.5 * fmax) and el2 <= max(2e-4, 2.5 * fl2), (case, prec, n, emax, el2, fmax, fl2)
            del ts3
        # graph step on a 4x longer clip (train.py:471-493)
        gclip = C.case_clip(case, opt, batch, 4 * frames, side, C.GRAPH_CLIP_SEED)
        with ops.deterministic():
            bp, GG = ts.graph_step(ts.to_device(gclip))
        for tag in ('o64', 'o32'):
            ref = refs[tag]
            for k, v in ref['GG'].items():
                assert abs(GG[k].item() - v) <= 1e-3 * max(1.0, abs(v)), (tag, k, GG[k].item(), v)
            assert C.out_errors(bp, ref['bp'])[0] <= 1e-3, tag
        # every state tensor after the three optimizer steps: buffers (spectral-norm u / v, running statistics) element-wise at
        # 1e-3 of their max; parameters through (abs-sum, L2) at 1e-3 + Adam's first-step floor lr per element (a parameter whose
        # true gradient is zero moves by +-lr in a rounding-determined direction)
        floor = 1.2 * opt.learning_rate
        for tag in ('o64', 'o32'):
            ref = refs[tag]
            bad_state, n_keys = [], 0
            for t, sd in (('G', ts.model.state_dict()), ('D', ts.discriminator.img_discriminator.state_dict())):
                assert set(sd) == set(k for (tt, k) in ref['state'] if tt == t)
                for k, v in sd.items():
                    rstat, n, rbuf = ref['state'][(t, k)]
                    n_keys += 1
                    if rbuf is not None:
                        err = (v.detach().cpu().double() - rbuf.double()).abs().max().item()
                        if err > _floor_scaled(case, 1e-3, 'state_buffers', 'max') * max(rbuf.abs().max().item(), 1e-6):
                            bad_state.append((k, 'buffer', err, rbuf.abs().max().item()))
                    gotst = C.stats(v.float())
                    for i, fl in ((1, floor * n), (2, floor * n ** 0.5)):
                        if abs(gotst[i] - rstat[i]) > 1e-3 * abs(rstat[i]) + fl:
                            bad_state.append((k, i, gotst[i], rstat[i]))
            # (CATER: 4 attributes => the reference's 492 + 56 keys; one attribute: AttributeEmbeddings has no mixing Linear,
            # attribute_embed.py:29-46 -- the key sets were compared with the oracle's above either way)
            assert n_keys == (492 + 56 if not case.startswith('c5') else len(ref['state'])) and n_keys >= 500
            assert not bad_state, (tag, len(bad_state), bad_state[:10])
        del ts
        # the default (non-deterministic) path, once: same weights, same clip, forward + losses.  Reported; gated on relative L2 and on
        # the losses only -- its max-norm varies from launch to launch with the order of the split-K atomics
        ts2 = TrainStep(opt, device, weights_seed=C.WEIGHTS_SEED)
        mo2, GL2, DL2 = ts2.gd_step(ts2.to_device(clip))
        for got2, rf in ((GL2, refs['o64']['GL']), (DL2, refs['o64']['DL'])):
            for k in rf:
                assert abs(got2[k].item() - rf[k]) <= 1e-3 * max(1.0, abs(rf[k])), ('default path', k, got2[k].item(), rf[k])
        _check_outputs(case, prec, mo2, refs, 'default path (spread report)', gate_max=False)
        # ... and its gradients + post-step buffers against o64 (the shipped default selects other kernels than the
        # deterministic mode -- the h16 halo family for the flow network, atomics instead of ordered slab reduces -- and was gated on
        # the forward only).  Same bounds as the deterministic step: the run-to-run spread of the atomics' order is inside them
        # (measured deviations are printed; deterministic mode: front <= 2.2e-2, decoder <= 4.4e-3, D <= 8.7e-3 at c3p).
        bad2, worst2 = _grad_report(ts2.model.named_parameters(), refs['o64']['G_grads'],
                                    lambda n: ('front', _grad_bound(case, 'front', 5e-2)) if C.is_front(n) else ('decoder', _grad_bound(case, 'decoder', 2e-2)),
                                    allow_missing='acts_to_boxes', per_elem=5e-6)
        # (per-element floor 5e-6 instead of the deterministic step's 2e-6: a convolution bias in front of a batch norm has an
        # analytically ZERO gradient -- what is stored is the rounding noise of a 10^5-term sum, and the atomics' order adds to it:
        # flows_network.res_flow.0.conv_0.bias measured 2.1e-6 per element)
        bad2d, worst2d = _grad_report(ts2.discriminator.img_discriminator.named_parameters(), refs['o64']['D_grads'], lambda n: ('D', _grad_bound(case, 'D', 2e-2)),
                                      per_elem=5e-6)
        worst2.update(worst2d)
        print('%s [%s] default path, worst gradient abs-sum deviations against o64:' % (case, prec), worst2)
        assert not bad2 and not bad2d, ('default path gradients', len(bad2), bad2[:10], bad2d[:10])
        bad_buf = []
        for t, sd in (('G', ts2.model.state_dict()), ('D', ts2.discriminator.img_discriminator.state_dict())):
            for k, v in sd.items():
                rbuf = refs['o64']['state'][(t, k)][2]
                if rbuf is None:
                    continue
                # (the oracle's state is the one after its graph step too; the graph net's buffers are none, so G / D buffers compare)
                err = (v.detach().cpu().double() - rbuf.double()).abs().max().item()
                if err > _floor_scaled(case, 1e-3, 'state_buffers', 'max') * max(rbuf.abs().max().item(), 1e-6):
                    bad_buf.append((k, err, rbuf.abs().max().item()))
        assert not bad_buf, ('default path post-step buffers', len(bad_buf), bad_buf[:10])
        det_vs_default = C.out_errors(mo2[0], mo[0])
        print('%s [%s] default path against deterministic path, imgs_pred (max-norm, relative L2): %.2e %.2e' % ((case, prec) + det_vs_default))
        del ts2
        return worst
    finally:
        ops.set_conv_precision(old)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c1_default_widths_matches_oracle(prec, device):
    _step_vs_oracle('c1', prec, device)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c1_bp_prev_default_widths_matches_oracle(prec, device):
    """c1 TRAINED with --bp_prev 1 at the default widths (reference generator.py:68-71; the oracle's path is pinned to the imported
    reference by tests/test_oracle_golden.py::test_oracle_matches_reference_bp_prev_training_step): the sequential generator under
    autograd -- the loss of frame t reaches the weights through frames t-1 and t-2 as well -- against o32 and o64, same gates as c1
    with the floors measured for THIS case (tests/golden/noise_floor.json 'c1bp')."""
    _step_vs_oracle('c1bp', prec, device)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c2prime_default_widths_matches_oracle(prec, device):
    """128 x 128 (c2's resolution), 3 frames, batch 1, default widths: full iteration against the oracle."""
    _step_vs_oracle('c2p', prec, device)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c3prime_default_widths_matches_oracle(prec, device):
    """256 x 256 (the benchmarked resolution), 4 frames, batch 1, default widths: full iteration against the oracle -- losses and
    outputs <= 1e-3, gradient statistics, all 548 post-step state keys."""
    _step_vs_oracle('c3p', prec, device)


def test_c3prime_flow_forward_margin(device):
    """How much of the output bound the flow network's forward arithmetic uses at 256 x 256: imgs_pred / flows_pred against the fp64
    oracle (deterministic mode) with that forward on (a) the fp16 3-term split -- the product's policy since round 3, must hold the step test's bound --, (b) the exact fp32
    tiles (rounds 1-2), (c) the bf16 3-term split (AG2V_FLOW_FWD diagnostic switch).  (b) and (c) are reported; (c) is asserted only to
    stay a small error (< 1e-2): it documents why the bf16 split is not used there."""
    import os
    from ag2video_amd import ops
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, batch = C.case_opt('c3p')
    errs = {}
    for label in ('h16', 'exact', 'bf16'):
        os.environ['AG2V_FLOW_FWD'] = label
        try:
            with ops.conv_precision('bf16x3'), ops.deterministic():
                ts = TrainStep(opt, device, weights_seed=1)
                shapes_G = {k: list(v.shape) for k, v in ts.model.state_dict().items()}
                shapes_D = {k: list(v.shape) for k, v in ts.discriminator.img_discriminator.state_dict().items()}
                ref = _oracle_steps('c3p', shapes_G, shapes_D)['o64']
                clip = ts.to_device(make_clip_batch(opt.vocab, batch, frames, (side, side), seed=1234))
                with torch.no_grad():
                    mo = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
            for i, name in ((0, 'imgs_pred'), (2, 'flows_pred')):
                errs[(label, name)] = C.out_errors(mo[i], ref['out'][i])[0]
            del ts
        finally:
            os.environ.pop('AG2V_FLOW_FWD', None)
    print('c3p flow-forward margin (error / max|ref|):', errs)
    assert errs[('h16', 'imgs_pred')] <= _img_bounds('c3p')[0], errs
    assert errs[('h16', 'flows_pred')] <= 2.0 * errs[('exact', 'flows_pred')] + 1e-6, errs       # fp32-class
    assert errs[('bf16', 'imgs_pred')] <= 1e-2, errs


# ---------------------------------------------------------------------------------------------------- c2 / c5 layer sizes
# (N, C, H, W, K, R, stride, pad): the GEMM shapes these configurations add over c3's (tests/test_fullsize_gpu.py):
LAYERS = {
    'c2_mlp_shared_128': (4, 512, 128, 128, 128, 3, 1, 1),
    'c2_up3_conv0_128': (4, 128, 128, 128, 64, 3, 1, 1),
    'c2_D_model1_65': (28, 64, 65, 65, 128, 4, 2, 2),          # discriminator, 4 x 7 frames, 4x4 stride 2
    'c5_mlp_shared_224': (2, 128, 224, 224, 128, 3, 1, 1),     # Something-Else: 128 layout channels
    'c5_gamma_beta_112': (2, 128, 112, 112, 256, 3, 1, 1),
    'c5_head_7': (2, 1024, 7, 7, 1024, 3, 1, 1),               # 7 x 7 latent map: 49 pixels per sample
    'c5_up0_28': (2, 1024, 28, 28, 512, 3, 1, 1),
    'c5_D_model0_img_224': (30, 3, 224, 224, 64, 4, 2, 2),     # image part of D's first convolution -> 113 x 113
    'c5_D_model2_57': (30, 128, 57, 57, 256, 4, 2, 2),
    'c5_Dtmp_model0_224': (6, 48, 224, 224, 64, 4, 2, 2),      # temporal discriminator (--temporal_D 1): a window of 16 frames x 3 channels
    'c5_Dtmp_model0_fpa4_224': (6, 12, 224, 224, 64, 4, 2, 2),   # ... of frames_per_action = 4 frames (the default width)
}


@pytest.mark.parametrize("name", sorted(LAYERS))
@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_conv_windows_and_adjoints_at_config_sizes(name, prec, device):
    """(a) windows of the full-size output (corners included: padding taps, odd sizes) against the oracle on the matching crop;
    (b) <conv(x, w), dy> == <x, dgrad(dy)> == <w, wgrad(dy)> in fp64 over the fp32 kernel outputs."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H, W, Ko, R, st, pad = LAYERS[name]
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        with ops.bf16x3_min_rows(48):
            x = _rand(N, C, H, W, seed=1, device=device).requires_grad_(True)
            w = _rand(Ko, C, R, R, seed=2, device=device, scale=(C * R * R) ** -0.5).requires_grad_(True)
            b = _rand(Ko, seed=3, device=device)
            y = ops.conv2d(x, w, b, st, pad)
            OH, OW = y.shape[2], y.shape[3]
            dy = _rand(N, Ko, OH, OW, seed=4, device=device)
            y.backward(dy)
        win = min(12, OH)
        for (n, oy, ox) in [(0, 0, 0), (N - 1, OH - win, OW - win), (N // 2, (OH - win) // 2, (OW - win) // 3)]:
            # input rows/cols feeding output window [oy, oy+win) x [ox, ox+win): [o*st - pad, (o+win-1)*st - pad + R)
            ys, xs = oy * st - pad, ox * st - pad
            ye, xe = (oy + win - 1) * st - pad + R, (ox + win - 1) * st - pad + R
            crop = x.detach()[n:n + 1, :, max(ys, 0):min(ye, H), max(xs, 0):min(xe, W)].cpu()
            crop = torch.nn.functional.pad(crop, (max(-xs, 0), max(xe - W, 0), max(-ys, 0), max(ye - H, 0)))
            ref = K.conv2d(crop, w.detach().cpu(), b.cpu(), st, 0)
            got = y.detach()[n:n + 1, :, oy:oy + win, ox:ox + win].cpu()
            assert ref.shape == got.shape, (ref.shape, got.shape)
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 2e-4, (name, prec, (n, oy, ox), err)
        yb = (y.detach() - b.view(1, -1, 1, 1)).double()
        a = (yb * dy.double()).sum().item()
        bx = (x.detach().double() * x.grad.double()).sum().item()
        bw = (w.detach().double() * w.grad.double()).sum().item()
        scale = (yb.pow(2).sum().sqrt() * dy.double().pow(2).sum().sqrt()).item()
        assert abs(a - bx) <= 2e-5 * scale, (name, a, bx, scale)
        assert abs(a - bw) <= 2e-5 * scale, (name, a, bw, scale)
    finally:
        ops.set_conv_precision(old)


def test_spade_norm_and_pyramid_at_224(device):
    """SPADE modulation moments and the segmentation-map pyramid (7, 14, ..., 224: factor-32 pyramid of a non-power-of-two side)."""
    from ag2video_amd import ops
    from oracle import kernels as K
    N, C, H = 2, 64, 224
    x = _rand(N, C, H, H, seed=4, device=device) * 3.0 + 1.5
    gb = torch.zeros(N, 2 * C, H, H, device=device)
    rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
    y = ops.spade_norm_fused(x, gb, rm, rv, True, 0.1, 1e-5, 1.0)
    m = y.double().mean(dim=(0, 2, 3))
    v = y.double().var(dim=(0, 2, 3), unbiased=False)
    assert m.abs().max().item() <= 1e-4 and (v - 1).abs().max().item() <= 1e-4
    seg = _rand(1, 16, H, H, seed=5)
    sizes = [(7 << i, 7 << i) for i in range(6)]
    sg = seg.clone().to(device).requires_grad_(True)
    sc = seg.clone().requires_grad_(True)
    got = ops.nearest_pyramid(sg, sizes)
    lg = lr = 0.0
    for i, ((h, w), yg) in enumerate(zip(sizes, got)):
        yr = K.nearest_resize(sc, h, w)
        assert torch.equal(yg.detach().cpu(), yr.detach()), (h, w)
        wg = _rand(*yr.shape, seed=20 + i)
        lg = lg + (yg * wg.to(device)).sum()
        lr = lr + (yr * wg).sum()
    lg.backward()
    lr.backward()
    assert (sg.grad.cpu() - sc.grad).abs().max().item() <= 1e-5 * sc.grad.abs().max().item()


def _finite_step(cfg, device, batch=None):
    import bench
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.engine import TrainStep
    opt, side, frames, b = bench.make_opt(cfg)
    if batch is not None:
        b = batch
        opt.batch_size = b
    opt.gpu_ids = [0]
    ts = TrainStep(opt, device, weights_seed=0)
    clip = ts.to_device(bench.make_clip(opt, cfg, b, frames, side, seed=1234))
    pg = dict(ts.model.named_parameters())
    name = next(k for k in pg if 'netG' in k and k.endswith('weight_orig'))
    p0 = pg[name].detach().clone()
    out, GL, DL = ts.gd_step(clip)
    assert set(GL) >= {'GAN_Img', 'GAN_Feat', 'loss_F_Warp', 'total_loss'} and set(DL) >= {'D_img_fake', 'D_img_real', 'total_img_loss'}
    assert all(bool(torch.isfinite(v).all()) for v in list(GL.values()) + list(DL.values())), (GL, DL)
    assert tuple(out[0].shape) == (b, frames, 3, side, side)
    assert bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[2]).all())
    assert (out[0][:, 0] == clip[0][:, 0]).all()                     # frame 0 is the ground-truth frame (generator.py:56)
    assert (pg[name].detach() - p0).abs().max().item() > 0
    gclip = ts.to_device(bench.make_clip(opt, cfg, b, frames, side, seed=4321))
    _, GG = ts.graph_step(gclip)
    assert bool(torch.isfinite(GG['total_loss']))
    return opt


def test_c2_step_runs_and_is_finite(device):
    """c2: CATER 128x128, 8 frames, batch 4, default widths."""
    _finite_step('c2', device)


@pytest.mark.parametrize("batch", [2, 8], ids=['batch2', 'batch8_as_benched'])
def test_c5_shaped_step_runs_and_is_finite(batch, device):
    """c5-shaped: Something-Else vocabulary (one attribute => semantic_nc 128, 2-3 objects incl. the hand), 224x224, 16 frames, at a
    short per-GPU batch and at the batch bench.py --config c5 runs (8: 120 frame passes per step, ~90 GB)."""
    torch.cuda.empty_cache()
    opt = _finite_step('c5', device, batch=batch)
    assert opt.semantic_nc == 128 and tuple(opt.image_size) == (224, 224)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
def test_step_c5prime_default_widths_matches_oracle(prec, device):
    """c5's geometry as a STEP against the oracle: Something-Else vocabulary, 224 x 224, 3 frames, batch 1, default
    widths, both arithmetics -- losses and outputs <= 1e-3, gradient statistics, every post-step state key (the same assertions as c3')."""
    _step_vs_oracle('c5p', prec, device)


def test_step_c5prime_temporal_discriminator_matches_oracle(device):
    """The same geometry with --temporal_D 1 --frames_per_action 4 (5 frames: one window of 4 generated frames through the
    reference's MultiscaleDiscriminatorT, discriminator.py:16-59): the temporal loss keys, and the temporal discriminator's gradients
    beside everything the c5' step checks."""
    _step_vs_oracle('c5pT', 'bf16x3', device)


@pytest.mark.parametrize("prec", ['bf16x3', 'fp32'])
@pytest.mark.parametrize("fixture", ['c2', 'c3h', 'c5h'])
def test_step_c2_full_size_matches_oracle_fixture(fixture, prec, device):
    """BASELINE.json's configs[1] at FULL size (CATER 128 x 128, 8 frames, batch 4, default widths): one whole iteration (G step, D step,
    graph step) on the GPU in deterministic mode against the CPU oracle's fp32 and fp64 runs of the same clip and weights, committed as
    tests/golden/oracle_c2.npz by tests/golden/make_oracle_c2.py (the oracle needs ~10 minutes for this step: too long to run inside a
    test, which is the only reason the c2 / c3 tests before round 6 were property checks).  Compared: every loss (1e-3 against both
    oracles), imgs_pred / flows_pred on the fixture's stride-4 pixel grid and boxes_pred in full, the abs-sum of every parameter's
    gradient, (abs-sum, L2) of every post-step state tensor.  Gates as in _step_vs_oracle, the floors being o32 against o64 on the very
    same statistic.  Second fixture, c3h: the benchmarked resolution with two samples per batch-norm statistic (256 x 256, 4 frames,
    batch 2; c3' is batch 1); third, c5h: c5's geometry (Something-Else vocabulary: one attribute => 128 layout channels, 224 x 224) at batch 2."""
    from ag2video_amd import ops
    from ag2video_amd.engine import TrainStep
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_%s.npz' % fixture)
    gold = np.load(path, allow_pickle=False)
    meta = json.loads(str(gold['meta']))
    case, st = meta['case'], int(meta['pixel_stride'])
    assert (meta['weights_seed'], meta['clip_seed'], meta['graph_clip_seed']) == (C.WEIGHTS_SEED, C.CLIP_SEED, C.GRAPH_CLIP_SEED)
    old = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        opt, side, frames, batch = C.case_opt(case)
        assert [side, frames, batch] == meta['side_frames_batch'] == list(C.FULL_CASES[fixture]) and case == fixture
        with ops.deterministic():
            ts = TrainStep(opt, device, weights_seed=C.WEIGHTS_SEED)
            clip = C.case_clip(case, opt, batch, frames, side, C.CLIP_SEED)
            mo, GL, DL = ts.gd_step(ts.to_device(clip))
        rep = {}
        for tag in ('o64', 'o32'):
            for grp, got in (('GL', GL), ('DL', DL)):
                keys = [k[len(tag) + len(grp) + 2:] for k in gold.files if k.startswith('%s.%s.' % (tag, grp))]
                assert set(keys) == set(got)
                for k in keys:
                    rf = float(gold['%s.%s.%s' % (tag, grp, k)])
                    assert abs(got[k].item() - rf) <= 1e-3 * max(1.0, abs(rf)), (tag, grp, k, got[k].item(), rf)

        def errs(a, b, scale=None):
            d = a.astype(np.float64) - b
            return float(np.abs(d).max() / (scale or np.abs(b).max())), float(np.linalg.norm(d) / np.linalg.norm(b))
        imgs = mo[0].detach().double().cpu().numpy()[..., ::st, ::st]
        flows = mo[2].detach().double().cpu().numpy()[..., ::st, ::st]
        amax = float(gold['o64.imgs_absmax'])
        f_img = errs(gold['o32.imgs_sub'], gold['o64.imgs_sub'], amax)          # the reference arithmetic's own distance from exact
        f_flow = errs(gold['o32.flows_sub'], gold['o64.flows_sub'])
        for tag, kmax in (('o64', 2.0), ('o32', 3.0)):
            e_img, e_flow = errs(imgs, gold[tag + '.imgs_sub'], amax), errs(flows, gold[tag + '.flows_sub'])
            e_box = errs(mo[1].detach().double().cpu().numpy(), gold[tag + '.boxes_pred'])
            rep[tag] = dict(imgs=e_img, flows=e_flow, boxes=e_box)
            assert e_img[1] <= (2.0 * f_img[1] if tag == 'o64' else 1e-3), (tag, 'imgs relative L2', e_img, f_img)
            assert e_img[0] <= max(1e-3, kmax * f_img[0]), (tag, 'imgs max-norm', e_img, f_img)
            assert e_flow[0] <= max(1e-3, 3.0 * f_flow[0]) and e_box[0] <= 1e-3, (tag, e_flow, e_box)
            assert np.mean(mo[3].cpu().numpy()[..., ::st, ::st] != gold[tag + '.conf_sub']) < 2e-3
        print('%s full size [%s]: (max-norm / max, relative L2) against the oracle fixture %s; the fp32 oracle itself: imgs %s flows %s'
              % (fixture, prec, rep, f_img, f_flow))
        # gradient abs-sums: teacher-forced bounds, never tighter than 2.5 x the fp32 oracle's own deviation on that parameter's group
        for gtag, params, groups in (('G_grads', ts.model.named_parameters(), None), ('D_grads', ts.discriminator.img_discriminator.named_parameters(), 'D')):
            rows = []
            for n, prm in params:
                key = 'o64.%s.%s' % (gtag, n)
                if key not in gold.files:
                    assert prm.grad is None or float(prm.grad.abs().sum()) == 0.0 or n.startswith('acts_to_boxes'), n
                    continue
                exact, o32v = float(gold[key][1]), float(gold['o32.%s.%s' % (gtag, n)][1])
                if exact <= 2e-6 * prm.numel():
                    continue
                grp = groups or ('front' if C.is_front(n) else 'decoder')
                rows.append((n, grp, abs(C.stats(prm.grad)[1] - exact) / exact, abs(o32v - exact) / exact))
            floor = {g: max(r[3] for r in rows if r[1] == g) for g in set(r[1] for r in rows)}
            base = {'front': 5e-2, 'decoder': 2e-2, 'D': 2e-2}
            bad = [r for r in rows if r[2] > max(base[r[1]], 2.5 * floor[r[1]])]
            print('%s full size [%s] %s: worst abs-sum deviation from o64 %s; the fp32 oracle itself %s'
                  % (fixture, prec, gtag, {g: max((r[2], r[0]) for r in rows if r[1] == g) for g in floor}, floor))
            assert len(rows) > (100 if gtag == 'G_grads' else 30) and not bad, (gtag, bad[:10])
        # the graph step and every state tensor after the three optimizer steps
        gclip = C.case_clip(case, opt, batch, 4 * frames, side, C.GRAPH_CLIP_SEED)
        with ops.deterministic():
            bp, GG = ts.graph_step(ts.to_device(gclip))
        for k, v in GG.items():
            rf = float(gold['o64.GG.' + k])
            assert abs(v.item() - rf) <= 1e-3 * max(1.0, abs(rf)), (k, v.item(), rf)
        assert errs(bp.detach().double().cpu().numpy(), gold['o64.graph_boxes'])[0] <= 1e-3
        floor_lr = 1.2 * opt.learning_rate
        bad_state, n_keys = [], 0
        for t, sd in (('G', ts.model.state_dict()), ('D', ts.discriminator.img_discriminator.state_dict())):
            for k, v in sd.items():
                ref = gold['o64.state.%s.%s' % (t, k)]
                n_keys += 1
                got, n = C.stats(v.float()), float(ref[3])
                for i, fl in ((1, floor_lr * n), (2, floor_lr * n ** 0.5)):
                    if abs(got[i] - ref[i]) > 1e-3 * abs(ref[i]) + fl:
                        bad_state.append((k, i, got[i], ref[i]))
        assert (n_keys == 492 + 56 or fixture.startswith('c5')) and n_keys >= 500 and not bad_state, (n_keys, len(bad_state), bad_state[:10])
        del ts
        # the SHIPPED default path at this size -- atomics, the co-scheduled graph-net programs on their side streams, the hoisted power
        # iterations (engine.TrainStep outside deterministic mode) -- against the same fixture: losses, imgs_pred in relative L2, gradient
        # abs-sums (per-element floor 5e-6: _step_vs_oracle's default-path leg)
        ts2 = TrainStep(opt, device, weights_seed=C.WEIGHTS_SEED)
        assert ts2.side_stream is not None
        mo2, GL2, DL2 = ts2.gd_step(ts2.to_device(clip))
        ts2.check_health()
        for grp, got in (('GL', GL2), ('DL', DL2)):
            for k, v in got.items():
                rf = float(gold['o64.%s.%s' % (grp, k)])
                assert abs(v.item() - rf) <= 1e-3 * max(1.0, abs(rf)), ('default path', grp, k, v.item(), rf)
        e2 = errs(mo2[0].detach().double().cpu().numpy()[..., ::st, ::st], gold['o64.imgs_sub'], amax)
        assert e2[1] <= 2.0 * f_img[1] and errs(mo2[1].detach().double().cpu().numpy(), gold['o64.boxes_pred'])[0] <= 1e-3, ('default path', e2, f_img)
        worst2 = {}
        for gtag, params, groups in (('G_grads', ts2.model.named_parameters(), None), ('D_grads', ts2.discriminator.img_discriminator.named_parameters(), 'D')):
            for n, prm in params:
                key = 'o64.%s.%s' % (gtag, n)
                if key not in gold.files or prm.grad is None:
                    continue
                exact, o32v = float(gold[key][1]), float(gold['o32.%s.%s' % (gtag, n)][1])
                grp = groups or ('front' if C.is_front(n) else 'decoder')
                dev_abs = abs(C.stats(prm.grad)[1] - exact)
                bound = max({'front': 5e-2, 'decoder': 2e-2, 'D': 2e-2}[grp] * exact, 2.5 * abs(o32v - exact)) + 5e-6 * prm.numel()
                if exact > 5e-6 * prm.numel():
                    worst2[grp] = max(worst2.get(grp, 0.0), dev_abs / exact)
                assert dev_abs <= bound, ('default path', gtag, n, dev_abs / max(exact, 1e-30))
        print('%s full size [%s] default path (side streams): imgs (max-norm, relative L2) %s, worst gradient abs-sum deviations %s' % (fixture, prec, e2, worst2))
        del ts2
    finally:
        ops.set_conv_precision(old)
