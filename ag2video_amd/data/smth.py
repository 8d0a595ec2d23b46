"""Something-Else annotations -> the tensors of the training step (reference: data/smth.py:20-390, data/SomethingElse/config.py,
data/dataset_params.py:117-152).

The data half of configuration c5: the reference reads a label table (``train.csv``: id, template, placeholders, nr_instances, err),
per-video box tracks (``offical_release_boxes/boxes_by_video_id/<id>/boxes.pkl``: one entry per frame with ``name``, ``nr_instances``
and ``labels`` = [{category, standard_category, gt_annotation, box2d}]), an object-name mapping
(``offical_release_boxes/objs_mapping.json``) and frame images (``frames/<name>``), and turns a video into
(frames, {'object': ids}, boxes, triplets, actions, id) -- the item ``ag2video_amd.data.cater.collate_fn`` pads into a batch.
Every tensor builder is pinned against the reference's own method on the same synthetic annotations (tests/golden/smth.npz), including its random draws (Python ``random`` for the spatial relations, ``numpy.random`` for the training window) and
its quirks (the relation test reads (x0, y0, w, h) rows as corner pairs).  ``write_synthetic_dataset`` produces a miniature
dataset of the same layout for tests and smoke runs.
"""
import ast
import csv
import json
import math
import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .cater import load_frames

# data/SomethingElse/config.py: the action vocabulary is this list (index = action id), and how many tracked instances
# (objects + the hand) a video of each template must have
ACTION_TEMPLATES = ('Putting [something] on a surface', 'Moving [something] up', 'Pushing [something] from left to right',
                    'Moving [something] down', 'Pushing [something] from right to left', 'Covering [something] with [something]',
                    'Uncovering [something]', 'Taking [one of many similar things on the table]', '__padding__')
INSTANCES_PER_TEMPLATE = {t: (3 if t.startswith('Covering') else 2) for t in ACTION_TEMPLATES[:-1]}
PREDICATES = ('__in_image__', 'right', 'above', 'below', 'left', 'surrounding', 'inside', 'cover', '__padding__')


def something_else_vocab(objs_mapping):
    """The vocabulary SmthElseDataset.__init__ builds (data/smth.py:85-104) from the object-name mapping."""
    names = ['__image__'] + sorted(set(objs_mapping.values()))
    v = {'action_idx_to_name': list(ACTION_TEMPLATES),
         'action_name_to_idx': {t: i for i, t in enumerate(ACTION_TEMPLATES)},
         'pred_name_to_idx': {p: i for i, p in enumerate(PREDICATES)},
         'attributes': {'object': {n: i for i, n in enumerate(names)}},
         'reverse_attributes': {'object': names}}
    v['pred_idx_to_name'] = {i: p for p, i in v['pred_name_to_idx'].items()}
    v['object_idx_to_name'] = names
    v['object_name_to_idx'] = v['attributes']['object']
    return v


def read_labels(path, objs_mapping):
    """Rows of the label table the reference keeps (data/smth.py:79-115, 361-372): no recorded error, a template of the vocabulary
    whose instance count matches the video's, every placeholder known to the object mapping.  -> [{'id', 'template',
    'placeholders' (mapped names), 'nr_instances'}] in file order."""
    rows = []
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            if 'err' in r and r['err'] not in ('', None):          # pandas reads an empty cell as NaN; anything else is "an error"
                continue
            t = r['template']
            if t not in INSTANCES_PER_TEMPLATE or INSTANCES_PER_TEMPLATE[t] != int(float(r['nr_instances'])):
                continue
            mapped = [objs_mapping.get(o) for o in ast.literal_eval(r['placeholders'])]
            if None in mapped:
                continue
            vid = r['id']
            rows.append({'id': int(vid) if vid.lstrip('-').isdigit() else vid, 'template': t, 'placeholders': mapped,
                         'nr_instances': int(float(r['nr_instances']))})
    return rows


def unique_frames(track):
    """The track without repeated frame names, first occurrence kept (data/smth.py:375-390)."""
    seen, out = set(), []
    for e in track:
        if e['name'] not in seen:
            seen.add(e['name'])
            out.append(e)
    return out


def action_window(track, n_instances, span, is_test, rng=np.random):
    """(f1, f2, s, e): the frames of the sample inside the stretch where all ``n_instances`` instances are annotated, and where in
    that stretch (as fractions) the sample starts / ends (data/smth.py:186-201).  Train: a random start (``rng.randint``)."""
    present = np.where(np.array([e['nr_instances'] for e in track]) == n_instances)[0]
    lo, hi = int(present.min()), int(present.max()) + 1
    f1 = lo
    if is_test:
        f2 = f1 + span
    else:
        if hi - span > lo:
            f1 = int(rng.randint(lo, hi - span))
        f2 = min(f1 + span, hi)
    return f1, f2, (f1 - lo + 1) / (hi - lo), (f2 - lo + 1) / (hi - lo)


def track_boxes(track, frame_size, n_instances, objs_mapping, vocab):
    """-> (boxes [F, n_instances, 4] as (x0, y0, w, h) / frame size, {'object': ids}) or None when the stretch shows another number
    of distinct instances (data/smth.py:224-263).  An instance is (standard_category, gt_annotation, mapped name), numbered by first
    appearance; a frame without a box for an instance repeats the previous frame's."""
    def key(lab):
        return (lab['standard_category'], lab['gt_annotation'], objs_mapping[lab['category']])
    slot = {}
    for e in track:
        for lab in e['labels']:
            slot.setdefault(key(lab), len(slot))
    out = np.zeros((len(track), n_instances, 4))
    for i, e in enumerate(track):
        out[i] = out[i - 1]
        for lab in e['labels']:
            b = lab['box2d']
            x1, x2, y1, y2 = b['x1'], b['x2'], b['y1'], b['y2']
            if x1 == x2 and y1 == y2:                            # degenerate annotation
                x1 = x2 = y1 = y2 = 0.0
            out[i, slot[key(lab)]] = x1, y1, x2 - x1, y2 - y1
    if len(slot) != n_instances:
        return None
    names = [k[-1] for k, _ in sorted(slot.items(), key=lambda kv: kv[1])]
    ids = torch.tensor([vocab['object_name_to_idx'][n] for n in names], dtype=torch.long)
    return torch.tensor(out / (tuple(frame_size) * 2), dtype=torch.float32), {'object': ids}


def hand_chain_actions(obj_ids, vocab, action_id, t_start, t_end):
    """[A, 5] rows [subject, action, object, t_start, t_end]: the hand (which must be the LAST instance, and the only hand) acts on
    object 0, object 0 on object 1, ... (data/smth.py:203-222); None when the instances do not have that form."""
    names = [vocab['object_idx_to_name'][int(i)] for i in obj_ids]
    if not names or names[-1] != 'hand' or 'hand' in names[:-1] or len(names) < 2:
        return None
    rows, prev = [], len(names) - 1
    for i in range(len(names) - 1):
        rows.append([prev, action_id, i, t_start, t_end])
        prev = i
    return torch.tensor(rows, dtype=torch.float32)


def spatial_triplets(boxes, vocab, include_relationships=True, rng=random):
    """[F, T, 3] (subject, predicate, object): per frame one spatial relation per real instance with a randomly chosen partner and
    direction (``rng.choice`` / ``rng.random``: Python's global generator in the reference), then (i, __in_image__, last row) for
    every instance (data/smth.py:135-184).  The relation test takes a box row as (x0, y0, x1, y1) although the rows hold
    (x0, y0, w, h) -- restated as it stands."""
    p2i = vocab['pred_name_to_idx']
    n = boxes.size(1) - 1
    frames = []
    for f in range(boxes.size(0)):
        rows = []
        for cur in range(n):
            others = [o for o in range(n) if o != cur]
            if not others or not include_relationships:
                break
            other = rng.choice(others)
            s, o = (cur, other) if rng.random() > 0.5 else (other, cur)
            sx0, sy0, sx1, sy1 = boxes[f][s]
            ox0, oy0, ox1, oy1 = boxes[f][o]
            dx = (sx0 + 0.5 * (sx1 - sx0)) - (ox0 + 0.5 * (ox1 - ox0))
            dy = (sy0 + 0.5 * (sy1 - sy0)) - (oy0 + 0.5 * (oy1 - oy0))
            theta = math.atan2(dy, dx)
            if sx0 < ox0 and sx1 > ox1 and sy0 < oy0 and sy1 > oy1:
                p = 'surrounding'
            elif sx0 > ox0 and sx1 < ox1 and sy0 > oy0 and sy1 < oy1:
                p = 'inside'
            elif theta >= 3 * math.pi / 4 or theta <= -3 * math.pi / 4:
                p = 'left'
            elif -3 * math.pi / 4 <= theta < -math.pi / 4:
                p = 'above'
            elif -math.pi / 4 <= theta < math.pi / 4:
                p = 'right'
            else:
                p = 'below'
            rows.append([s, p2i[p], o])
        rows += [[i, p2i['__in_image__'], n] for i in range(n)]
        frames.append(rows)
    return torch.tensor(frames, dtype=torch.long)


class SmthElseScenes(Dataset):
    """Counterpart of the reference's SmthElseDataset (data/smth.py:20-358) over ``data_root/{offical_release_boxes/..., frames/...}``
    and a label table.  An item is (frames [F, 3, H, W], {'object': ids [O]}, boxes [F, O, 4], triplets [F, T, 3], actions [A, 7],
    '<id>_<f1>-<f2>'); a video the reference rejects (instances hidden, too short, unreadable) yields an all-None item, which
    collate_fn drops."""

    def __init__(self, data_root, labels, is_test=False, is_val=False, image_size=(64, 64), frames_per_action=16,
                 initial_frames_per_sample=16, include_relationships=True, max_samples=None):
        self.root, self.is_test, self.is_val = data_root, is_test, is_val
        self.image_size, self.frames_per_action, self.span = tuple(image_size), frames_per_action, initial_frames_per_sample
        self.include_relationships, self.max_samples = include_relationships, max_samples
        with open(os.path.join(data_root, 'offical_release_boxes', 'objs_mapping.json')) as f:
            self.objs_mapping = json.load(f)
        self.vocab = something_else_vocab(self.objs_mapping)
        self.rows = read_labels(labels, self.objs_mapping)

    def __len__(self):
        return len(self.rows) if self.max_samples is None else min(len(self.rows), self.max_samples)

    def track(self, video_id):
        with open(os.path.join(self.root, 'offical_release_boxes', 'boxes_by_video_id', str(video_id), 'boxes.pkl'), 'rb') as f:
            return unique_frames(pickle.load(f))        # (the dataset's own annotation files: as trusted as the reference trusts them)

    def __getitem__(self, index):
        fail = (None,) * 6
        row = self.rows[index]
        try:
            track = self.track(row['id'])
            paths = sorted(os.path.join(self.root, 'frames', e['name']) for e in track)
            n_inst = INSTANCES_PER_TEMPLATE[row['template']]
            f1, f2, t0, t1 = action_window(track, n_inst, self.span, self.is_test)
            if not self.is_val and (f2 - f1) < min(self.span, 8):
                return fail
            frames, track = list(range(f1, f2)), track[f1:f2]
            per_action, span = (len(frames), len(frames)) if self.is_test else (self.frames_per_action, self.span)
            frames, track = frames[0:span:span // per_action], track[0:span:span // per_action]
            if len(frames) < per_action:                          # a short stretch is padded with its last frame
                pad = per_action - len(frames)
                frames, track = frames + frames[-1:] * pad, track + track[-1:] * pad
            from PIL import Image
            with Image.open(paths[0]) as im:
                frame_size = im.size
            got = track_boxes(track, frame_size, n_inst, self.objs_mapping, self.vocab)
            if got is None:
                return fail
            boxes, objs = got
            acts = hand_chain_actions(objs['object'], self.vocab, self.vocab['action_name_to_idx'][row['template']], t0, t1)
            if acts is None:
                return fail
            acts = torch.cat([acts, torch.zeros(acts.size(0), 2)], dim=1)
            triplets = spatial_triplets(boxes, self.vocab, self.include_relationships)
            vids = load_frames([paths[f] for f in frames], self.image_size)
            return vids, objs, boxes, triplets, acts, '%s_%d-%d' % (row['id'], f1, f2)
        except Exception as e:                                    # the reference reports and skips the video as well
            print("Error: in video_id {} with {}".format(row['id'], e))
            return fail


def dataset_vocab(data_root):
    """The vocabulary of the dataset under data_root (its object names come from the dataset's own mapping file)."""
    with open(os.path.join(data_root, 'offical_release_boxes', 'objs_mapping.json')) as f:
        return something_else_vocab(json.load(f))


def build_loaders(args, vocab):
    """(train_loader, train_graph_loader) over ``args.data_root`` with the clip shapes of the reference's configuration table
    (data/dataset_params.py:117-135): frames_per_action consecutive frames for the G/D step, 4 * frames_per_action_graph for the
    graph step; label table = --train_list or <data_root>/train.csv; batch_size, shuffle, drop_last as scripts/train.py:116-134."""
    from functools import partial
    from torch.utils.data import DataLoader
    from .cater import collate_fn
    labels = getattr(args, 'train_list', None) or os.path.join(args.data_root, 'train.csv')
    mk = lambda f: SmthElseScenes(args.data_root, labels, image_size=args.image_size, frames_per_action=f, initial_frames_per_sample=f)
    kw = dict(batch_size=args.batch_size, shuffle=True, drop_last=True, num_workers=args.loader_num_workers,
              collate_fn=partial(collate_fn, vocab))
    return DataLoader(mk(args.frames_per_action), **kw), DataLoader(mk(4 * args.frames_per_action_graph), **kw)


# ------------------------------------------------------------------------------------------------ synthetic dataset files
_SYN_OBJECTS = {'a bottle': 'bottle', 'bottle': 'bottle', 'cup': 'cup', 'a cloth': 'cloth', 'cloth': 'cloth', 'pen': 'pen',
                'hand': 'hand', 'box': 'box'}


def synthetic_track(seed, template, n_frames=24, frame_size=(120, 90), lead=2, tail=2, hide=None):
    """A box track with the schema of ``boxes.pkl``: ``lead`` / ``tail`` frames where the hand is not yet / no longer annotated, every
    instance drifting linearly; ``hide`` = (instance, frame) drops one box of one frame (the reader repeats the previous one)."""
    rs = np.random.RandomState(500 + seed)
    n_inst = INSTANCES_PER_TEMPLATE[template]
    cats = [str(c) for c in rs.choice(['a bottle', 'cup', 'a cloth', 'pen', 'box'], n_inst - 1, replace=False)] + ['hand']
    W, H = frame_size
    p0 = rs.uniform(0.1, 0.6, (n_inst, 2)) * [W, H]
    v = rs.uniform(-0.4, 0.4, (n_inst, 2)) * [W, H] / n_frames
    wh = rs.uniform(0.12, 0.3, (n_inst, 2)) * [W, H]
    track = []
    for f in range(n_frames):
        full = lead <= f < n_frames - tail
        labs = []
        for i in range(n_inst):
            if (i == n_inst - 1 and not full) or (hide is not None and hide == (i, f)):
                continue
            x1, y1 = p0[i] + v[i] * f
            labs.append({'category': cats[i], 'standard_category': 'hand' if cats[i] == 'hand' else '%04d' % i,
                         'gt_annotation': 'hand' if cats[i] == 'hand' else 'object %d' % i,
                         'box2d': {'x1': float(x1), 'x2': float(x1 + wh[i, 0]), 'y1': float(y1), 'y2': float(y1 + wh[i, 1])}})
        track.append({'name': 'v%d/%04d.png' % (seed, f + 1), 'nr_instances': len(labs), 'labels': labs})
    return track, cats


def write_synthetic_dataset(root, n_videos=4, n_frames=24, frame_size=(120, 90), seed=0):
    """data_root + label table of the reference's layout with random frames: -> (data_root, labels.csv)."""
    from PIL import Image
    os.makedirs(os.path.join(root, 'offical_release_boxes'), exist_ok=True)
    with open(os.path.join(root, 'offical_release_boxes', 'objs_mapping.json'), 'w') as f:
        json.dump(_SYN_OBJECTS, f)
    rs = np.random.RandomState(seed)
    lines = [['', 'id', 'label', 'template', 'placeholders', 'nr_instances', 'err']]
    for k in range(n_videos):
        template = ACTION_TEMPLATES[(seed + k) % (len(ACTION_TEMPLATES) - 1)]
        vid = 1000 + seed * 100 + k
        track, cats = synthetic_track(vid, template, n_frames, frame_size)
        d = os.path.join(root, 'offical_release_boxes', 'boxes_by_video_id', str(vid))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'boxes.pkl'), 'wb') as f:
            pickle.dump(track, f)
        for e in track:
            p = os.path.join(root, 'frames', e['name'])
            os.makedirs(os.path.dirname(p), exist_ok=True)
            Image.fromarray(rs.randint(0, 256, (frame_size[1], frame_size[0], 3), dtype=np.uint8)).save(p)
        lines.append([str(k), str(vid), template.lower(), template, repr(cats[:-1]), str(INSTANCES_PER_TEMPLATE[template]), ''])
    labels = os.path.join(root, 'labels.csv')
    with open(labels, 'w', newline='') as f:
        csv.writer(f).writerows(lines)
    return root, labels
