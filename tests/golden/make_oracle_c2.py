#!/usr/bin/env python
"""BASELINE.json's configs[1] at FULL size -- CATER 128 x 128, 8 frames, batch 4, default widths -- through the CPU oracle, once in fp32
(o32: the restatement of the reference arithmetic, pinned to the imported reference by tests/test_oracle_golden.py) and once in fp64 (o64:
the exact result of the same graph on the same fp32 weights and clip).  The oracle needs ~4 + ~8 minutes and ~25 GB for this step, far
beyond what a test may spend, so its results are committed as a fixture: tests/golden/oracle_c2.npz.  CPU only; nothing from
/root/reference is read (the oracle is the checker the GPU tests use everywhere else, here run ahead of time).

    python tests/golden/make_oracle_c2.py [c2 | c3h]          # -> tests/golden/oracle_<case>.npz

c3h (round 6): the benchmarked resolution with MORE THAN ONE sample per batch-norm statistic -- 256 x 256, 4 frames, batch 2 (c3' is batch 1;
the full c3 step would need ~150 GB and ~1.5 h on the CPU).

Stored per oracle (o32 / o64): every loss of the G, D and graph steps; imgs_pred and flows_pred on a stride-4 pixel grid
([4, 8, C, 32, 32]: 1/16 of the pixels) and boxes_pred in full; (sum, abs-sum, L2) of every parameter's gradient; (sum, abs-sum, L2) of
every state tensor after the three optimizer steps.  The o32 copy is what makes the o64 comparison meaningful: o32 vs o64 on the same
statistic is the reference arithmetic's own distance from the exact result (tests/test_configs_gpu.py derives its gates from it, as
it does from noise_floor.json for the cases whose oracle runs inside the test).
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

STRIDE = 4


def main():
    CASE = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    nthr = min(16, os.cpu_count() or 1)
    torch.set_num_threads(nthr)
    sG, sD, sT = C.cpu_key_shapes(CASE)
    out = {'meta': np.array(json.dumps({'case': CASE, 'side_frames_batch': list(C.FULL_CASES[CASE]), 'weights_seed': C.WEIGHTS_SEED,
                                        'clip_seed': C.CLIP_SEED, 'graph_clip_seed': C.GRAPH_CLIP_SEED, 'pixel_stride': STRIDE,
                                        'threads': nthr, 'torch': torch.__version__}))}
    for tag, dt in (('o32', torch.float32), ('o64', torch.float64)):
        t0 = time.time()
        r = C.oracle_step(CASE, sG, sD, sT, dt)
        print('%s: %.0f s' % (tag, time.time() - t0), {k: round(v, 5) for k, v in r['GL'].items()}, flush=True)
        for grp in ('GL', 'DL', 'GG'):
            for k, v in r[grp].items():
                out['%s.%s.%s' % (tag, grp, k)] = np.float64(v)
        imgs, boxes, flows, conf = r['out']
        out[tag + '.imgs_sub'] = imgs[..., ::STRIDE, ::STRIDE].double().numpy()
        out[tag + '.flows_sub'] = flows[..., ::STRIDE, ::STRIDE].double().numpy()
        out[tag + '.conf_sub'] = conf[..., ::STRIDE, ::STRIDE].float().numpy()
        out[tag + '.boxes_pred'] = boxes.double().numpy()
        out[tag + '.imgs_absmax'] = np.float64(imgs.abs().max().item())
        out[tag + '.graph_boxes'] = r['bp'].double().numpy()
        for g in ('G_grads', 'D_grads'):
            for n, st in r[g].items():
                if st is not None:
                    out['%s.%s.%s' % (tag, g, n)] = st
        for (t, k), (st, numel, buf) in r['state'].items():
            out['%s.state.%s.%s' % (tag, t, k)] = np.concatenate([st, [numel]])
        del r
    path = os.path.join(HERE, 'oracle_%s.npz' % CASE)
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f MB' % (path, len(out), os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    main()
