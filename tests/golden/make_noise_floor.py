#!/usr/bin/env python
"""Measures the fp32 ORACLE's own rounding against the same iteration in fp64 (oracle/step.py dtype=torch.float64: the same fp32
weights and the same fp32 clip, widened) for every default-width step case of tests/_cases.py, and writes
tests/golden/noise_floor.json.  CPU only; runs anywhere the repo runs (nothing from /root/reference is needed: the oracle is pinned
to the reference by tests/test_oracle_golden.py):

    python tests/golden/make_noise_floor.py [case ...]          # default: every case; merges into the existing file

Why: the GPU step tests compare maxima over 10^5..10^6 pixels and sums over 10^3..10^7 gradient elements with the CPU oracle.  A bound
on such a statistic means something only next to the distance of the fp32 oracle itself from the exact result of the same graph.  Per
case the file holds, for o32 (fp32 oracle) against o64:
  out.<name>.{max,l2}   max|o32-o64| / max|o64| and ||o32-o64|| / ||o64||  for imgs_pred / boxes_pred / flows_pred, conf mismatch rate
  loss.<key>            |o32-o64| / max(1,|o64|)
  grad.<group>.{abssum_dev, l2} worst over the group's parameters of |S(o32)-S(o64)| / S(o64) with S = sum|g| (the statistic the GPU
                        test bounds) and of ||g32-g64|| / ||g64||; groups: front (flow network + appearance graph net), decoder (the
                        rest of G), D, T (temporal D); `worst` names the parameter
  threads_1_vs_n        the same figures for o32 at 1 thread against o32 at N threads (another summation order of the SAME arithmetic)
  imgs_pred_o32_samples_vs_o64   imgs_pred {max, l2} of o32 at 1 / 2 / 4 / N threads, each against o64: samples of the fp32 oracle's own
                        distance from the exact result -- the max-norm is a maximum over 10^5-10^6 pixels of a heavy-tailed error field
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _cases as C  # noqa: E402

OUT = os.path.join(HERE, 'noise_floor.json')


def compare(a, b, tiny_per_elem=2e-6):
    """a against reference b (both results of C.oracle_step(keep_grads=True))."""
    r = {'out': {}, 'loss': {}, 'grad': {}}
    for i, name in enumerate(('imgs_pred', 'boxes_pred', 'flows_pred')):
        mx, l2 = C.out_errors(a['out'][i], b['out'][i])
        r['out'][name] = {'max': mx, 'l2': l2}
    r['out']['conf_mismatch'] = float((a['out'][3].double() != b['out'][3].double()).double().mean())
    for grp in ('GL', 'DL', 'GG'):
        for k, v in b[grp].items():
            r['loss'][grp + '.' + k] = abs(a[grp][k] - v) / max(1.0, abs(v))
    mx, l2 = C.out_errors(a['bp'], b['bp'])
    r['out']['graph_step_boxes'] = {'max': mx, 'l2': l2}
    groups = {}
    for tag in ('G_grads', 'D_grads', 'T_grads'):
        if tag not in b:
            continue
        for n, gb in b[tag].items():
            ga = a[tag].get(n)
            if gb is None or ga is None:
                continue
            grp = ('front' if C.is_front(n) else 'decoder') if tag == 'G_grads' else tag[0]
            sb, sa = gb.double().abs().sum().item(), ga.double().abs().sum().item()
            if sb <= tiny_per_elem * gb.numel():          # the GPU test's absolute floor: a (numerically) zero gradient
                continue
            dev = abs(sa - sb) / sb
            l2 = float((ga.double() - gb.double()).pow(2).sum().sqrt() / gb.double().pow(2).sum().sqrt())
            g = groups.setdefault(grp, {'abssum_dev': 0.0, 'abssum_worst': '', 'l2': 0.0, 'l2_worst': '', 'n': 0})
            g['n'] += 1
            if dev > g['abssum_dev']:
                g['abssum_dev'], g['abssum_worst'] = dev, n
            if l2 > g['l2']:
                g['l2'], g['l2_worst'] = l2, n
    r['grad'] = groups
    worst_buf = (0.0, '')
    for key, (st, n, buf) in b['state'].items():
        if buf is None:
            continue
        abuf = a['state'][key][2]
        e = float((abuf.double() - buf.double()).abs().max() / max(buf.double().abs().max().item(), 1e-6))
        if e > worst_buf[0]:
            worst_buf = (e, '%s.%s' % key)
    r['state_buffers'] = {'max': worst_buf[0], 'worst': worst_buf[1]}
    return r


def vgg_floor():
    """The VGG-loss gradient of tests/golden/extras.npz (reference golden, fp32) against the oracle's same loss in fp64: d|a - b| is a
    sign function, so a feature pair closer than the arithmetic's rounding flips its whole term (tests/test_extras_gpu.py)."""
    from oracle import model as M
    gold = np.load(os.path.join(HERE, 'extras.npz'), allow_pickle=False)
    shapes = json.loads(str(gold['vgg.keys']))
    row = {}
    grads = {}
    for tag, dt in (('o32', None), ('o64', torch.float64)):
        V = {k: v.detach() for k, v in M.make_state(shapes, 3, 'V.', dt).items()}
        x = torch.from_numpy(gold['vgg.x']).to(dt or torch.float32).requires_grad_(True)
        y = torch.from_numpy(gold['vgg.y']).to(dt or torch.float32)
        loss = M.vgg_loss(V, x, y)
        loss.backward()
        grads[tag] = x.grad.double().numpy()
        row[tag + '_loss'] = loss.item()
    grads['reference_golden'] = gold['vgg.grad_x'].astype(np.float64)
    r64 = grads['o64']
    for tag in ('o32', 'reference_golden'):
        d = np.abs(grads[tag] - r64)
        row[tag + '_vs_o64'] = {'max_of_max': float(d.max() / np.abs(r64).max()), 'mean_of_mean': float(d.mean() / np.abs(r64).mean()),
                                'l2': float(np.linalg.norm(d) / np.linalg.norm(r64))}
    return row


def main():
    cases = sys.argv[1:] or (list(C.STEP_CASES) + ['vgg'])
    nthr = min(16, os.cpu_count() or 1)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if 'vgg' in cases:
        cases.remove('vgg')
        res['vgg_grad'] = vgg_floor()
        print('vgg', json.dumps(res['vgg_grad']), flush=True)
        json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)
    for case in cases:
        sG, sD, sT = C.cpu_key_shapes(case)
        row = {'side_frames_batch': list(C.STEP_CASES[case]), 'threads': nthr, 'torch': torch.__version__}
        torch.set_num_threads(nthr)
        t0 = time.time()
        o32 = C.oracle_step(case, sG, sD, sT, torch.float32, keep_grads=True)
        row['o32_s'] = round(time.time() - t0, 1)
        t0 = time.time()
        o64 = C.oracle_step(case, sG, sD, sT, torch.float64, keep_grads=True)
        row['o64_s'] = round(time.time() - t0, 1)
        row['o32_vs_o64'] = compare(o32, o64)
        print(case, json.dumps(row['o32_vs_o64']['out']), json.dumps(row['o32_vs_o64']['grad']), flush=True)
        torch.set_num_threads(1)
        o32_1 = C.oracle_step(case, sG, sD, sT, torch.float32, keep_grads=True)
        row['threads_1_vs_n'] = compare(o32_1, o32)
        print(case, 'threads', json.dumps(row['threads_1_vs_n']['out']), flush=True)
        # the fp32 oracle's OWN spread: other summation orders of the same arithmetic (thread counts), each against o64
        samples = {str(nthr): row['o32_vs_o64']['out']['imgs_pred'], '1': compare(o32_1, o64)['out']['imgs_pred']}
        del o32, o32_1
        for t in (2, 4):
            if t < nthr:
                torch.set_num_threads(t)
                o = C.oracle_step(case, sG, sD, sT, torch.float32)
                mx, l2 = C.out_errors(o['out'][0], o64['out'][0])
                samples[str(t)] = {'max': mx, 'l2': l2}
        torch.set_num_threads(nthr)
        row['imgs_pred_o32_samples_vs_o64'] = samples
        print(case, 'samples', json.dumps(samples), flush=True)
        del o64
        res[case] = row
        json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
