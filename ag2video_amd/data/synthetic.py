"""Seeded CATER-shaped synthetic clips in the exact padded layout of the reference
``collate_fn`` (data/dataset_params.py:8-104): objs pad 0, boxes pad -1, triplets pad
``[0, __padding__, 0]``, actions pad ``[0, __padding__, 0, 0, 0, 0, 0]``.

Shapes/ranges follow SURVEY.md 8(d) (data/cater.py:101-106,161-178,250-327,446-466).
NumPy RandomState => identical bytes on every machine.
"""
import numpy as np
import torch


def make_clip_batch(vocab, batch_size, frames, image_size, seed=1234, min_objs=5, max_objs=10,
                    max_actions=6, pad_objects_to=None):
    """Returns (imgs[B,F,3,H,W] f32, objs[B,O,A] i64, boxes[B,F,O,4] f32 xywh,
    triplets[B,F,T,3] i64, actions[B,Aact,7] f32, video_id list) on the CPU."""
    rs = np.random.RandomState(seed)
    H, W = image_size
    attrs = list(vocab['attributes'].keys())
    n_attr = len(attrs)
    pad_pred = vocab['pred_name_to_idx']['__padding__']
    pad_act = vocab['action_name_to_idx']['__padding__']
    n_act_cls = pad_act  # classes 1..pad_act-1 are real actions
    per = []
    for b in range(batch_size):
        n = int(rs.randint(min_objs, max_objs + 1))
        O = n + 1  # + the __image__ row (all-zero attributes, box [0,0,1,1]) data/cater.py:161,327
        objs = np.zeros((O, n_attr), np.int64)
        for k, a in enumerate(attrs):
            hi = max(vocab['attributes'][a].values())
            objs[:n, k] = rs.randint(1, hi + 1, size=n)
        bw = rs.uniform(0.094, 0.22, size=n)
        bh = rs.uniform(0.125, 0.29, size=n)
        x0 = rs.uniform(0.0, 1.0 - bw)
        y0 = rs.uniform(0.0, 1.0 - bh)
        boxes = np.zeros((frames, O, 4), np.float32)
        for t in range(frames):
            if t > 0:
                x0 = np.clip(x0 + rs.uniform(-0.03, 0.03, size=n), 0.0, 1.0 - bw)
                y0 = np.clip(y0 + rs.uniform(-0.03, 0.03, size=n), 0.0, 1.0 - bh)
            boxes[t, :n, 0], boxes[t, :n, 1], boxes[t, :n, 2], boxes[t, :n, 3] = x0, y0, bw, bh
            boxes[t, n] = (0.0, 0.0, 1.0, 1.0)
        trip = np.zeros((frames, n, 3), np.int64)
        trip[:, :, 0] = np.arange(n)[None]
        trip[:, :, 1] = vocab['pred_name_to_idx']['__in_image__']
        trip[:, :, 2] = n
        A = int(rs.randint(1, max_actions + 1))
        acts = np.zeros((A, 7), np.float32)
        for i in range(A):
            s = rs.randint(0, n)
            a = rs.randint(1, n_act_cls)
            o = s if rs.rand() < 0.7 else rs.randint(0, n)
            t1 = rs.uniform(-1.0, 0.5)
            t2 = rs.uniform(0.5, 2.0)
            name = vocab['action_idx_to_name'][a]
            if name in ('_slide', '_pick_place') or 'attributes' in vocab and len(attrs) == 1:
                xe, ye = rs.uniform(0.05, 0.8), rs.uniform(0.05, 0.8)
            else:
                xe, ye = 0.0, 0.0
            acts[i] = (s, a, o, t1, t2, xe, ye)
        per.append((objs, boxes, trip, acts))
    O_max = max(p[0].shape[0] for p in per)
    if pad_objects_to is not None:
        O_max = max(O_max, pad_objects_to)
    T_max = max(p[2].shape[1] for p in per)
    A_max = max(p[3].shape[0] for p in per)
    objs_b = np.zeros((batch_size, O_max, n_attr), np.int64)
    boxes_b = -np.ones((batch_size, frames, O_max, 4), np.float32)
    trip_b = np.zeros((batch_size, frames, T_max, 3), np.int64)
    trip_b[..., 1] = pad_pred
    acts_b = np.zeros((batch_size, A_max, 7), np.float32)
    acts_b[..., 1] = pad_act
    for b, (objs, boxes, trip, acts) in enumerate(per):
        objs_b[b, :objs.shape[0]] = objs
        boxes_b[b, :, :boxes.shape[1]] = boxes
        trip_b[b, :, :trip.shape[1]] = trip
        acts_b[b, :acts.shape[0]] = acts
    imgs = np.clip(rs.standard_normal((batch_size, frames, 3, H, W)), -2.2, 2.7).astype(np.float32)
    vids = ['synthetic_%06d' % (seed * 1000 + b) for b in range(batch_size)]
    return (torch.from_numpy(imgs), torch.from_numpy(objs_b), torch.from_numpy(boxes_b),
            torch.from_numpy(trip_b), torch.from_numpy(acts_b), vids)
