"""CATER scene files -> the tensors of the training step (reference: data/cater.py:157-466, data/dataset_params.py:8-104).

SURVEY.md 8(f) row 4 ("synthetic -> real data loaders"): the reference's dataset reads CATER scene-graph JSON
(``scenes/<video>.json``: objects with per-frame 3-D locations + a ``movements`` table) and cached frames
(``videos/<video>/00000.png ...``, which it extracts from the .avi with ffmpeg on first use -- ffmpeg / skvideo are not installed
here, so only the cache layout is read) and turns them into (frames, objs, boxes, triplets, actions, id) items that ``collate_fn`` pads
into a batch.  Every tensor builder below is pinned against the reference's own method on the same scene (tests/golden/extras.npz).
``synthetic_scene`` / ``write_synthetic_dataset`` produce scene files of the same schema for tests and smoke runs.
"""
import glob
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

# camera of CATER's fixed-camera renders (data/cater.py:339-343)
CATER_CAM = np.array([[1.4503, 1.6376, 0.0000, -0.0251],
                      [-1.0346, 0.9163, 2.5685, 0.0095],
                      [-0.6606, 0.5850, -0.4748, 10.5666],
                      [-0.6592, 0.5839, -0.4738, 10.7452]])
FRAME_W, FRAME_H = 320, 240
# half extents in pixels of the 320 x 240 frame: (half width, extent above the centre, extent below) per (shape, size)
# (data/cater.py:250-313)
_EXTENT = {'large': (35, 35, 35), 'medium': (25, 25, 25), 'small': (15, 15, 15)}
BOX_EXTENTS = {shape: dict(_EXTENT) for shape in ('spl', 'cylinder', 'cube', 'sphere', 'cone')}
BOX_EXTENTS['cone'] = {'large': (35, 25, 40), 'medium': (25, 15, 30), 'small': (20, 20, 20)}
BOX_EXTENTS['sphere'] = {'large': (35, 25, 40), 'medium': (25, 25, 25), 'small': (15, 15, 15)}
IMG_MEAN, IMG_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
MIN_ACTION_FRAMES = 12          # shorter movements are dropped (data/cater.py:195-196)


def project_points(pts):
    """[N, 3] world points -> [N, 2] image coordinates in [-1, 1], y pointing down (data/cater.py:331-351)."""
    h = np.hstack([np.asarray(pts, np.float64), np.ones((len(pts), 1))]) @ CATER_CAM.T
    return np.stack([h[:, 0] / h[:, 3], -h[:, 1] / h[:, 3]], axis=1)


def scene_objects(sg, vocab):
    """{attribute: LongTensor [O + 1]}: the vocabulary ids of every object's attributes + the __image__ row (data/cater.py:157-163)."""
    out = {}
    for attr, table in vocab['attributes'].items():
        out[attr] = torch.tensor([table[o[attr]] for o in sg['objects']] + [table['__image__']], dtype=torch.long)
    return out


def scene_boxes(sg):
    """[n_frames, O + 1, 4] boxes (x0, y0, w, h) in [0, 1] for every frame of the video; last row = the whole image
    (data/cater.py:234-329)."""
    per_obj = []
    for o in sg['objects']:
        xy = project_points(np.array(list(o['locations'].values())))
        cx, cy = (xy[:, 0] + 1) * FRAME_W / 2, (xy[:, 1] + 1) * FRAME_H / 2
        hw, up, down = BOX_EXTENTS[o['shape']][o['size']]
        per_obj.append(np.stack([(cx - hw) / FRAME_W, (cy - up) / FRAME_H, np.full_like(cx, 2.0 * hw / FRAME_W),
                                 np.full_like(cy, float(up + down) / FRAME_H)], axis=1))
    n = len(per_obj[0])
    per_obj.append(np.tile([[0.0, 0.0, 1.0, 1.0]], (n, 1)))
    return torch.tensor(np.stack(per_obj, axis=1), dtype=torch.float32)


def frame_triplets(n_frames, n_objects, vocab):
    """[F, O, 3]: (object i, __in_image__, image row) for every object and frame (data/cater.py:165-178)."""
    row = [[i, vocab['pred_name_to_idx']['__in_image__'], n_objects] for i in range(n_objects)]
    return torch.tensor([row] * n_frames, dtype=torch.long)


def scene_actions(sg, vocab):
    """All movements lasting >= MIN_ACTION_FRAMES as rows [subject, action id, object, first frame, last frame]."""
    index = {o['instance']: i for i, o in enumerate(sg['objects'])}
    rows = []
    for name, moves in sg['movements'].items():
        for action, other, f0, f1 in moves:
            if f1 - f0 < MIN_ACTION_FRAMES:
                continue
            s = index[name]
            rows.append([s, vocab['action_name_to_idx'][action], index[other] if other is not None else s, f0, f1])
    return rows


def choose_window(rows, max_frame, span, is_test, rng=np.random):
    """Frame window [start, end) of a sample and the actions overlapping it (data/cater.py:180-212): test = from the first action
    on; train = a random window of ``span`` frames (drawn with ``rng.randint``, the reference uses numpy's global state)."""
    starts, ends = [r[3] for r in rows], [r[4] for r in rows]
    if is_test:
        a = min(starts)
        b = min(max(ends), a + span)
    else:
        a = rng.randint(0, min(max(ends), max_frame) - span + 1)
        b = a + span
    keep = [r for r in rows if not (r[3] > b or r[4] < a)]
    return torch.tensor(keep, dtype=torch.long), (a, b)


def normalize_actions(actions, boxes_all, start, end, vocab):
    """[A, 5] integer rows -> [A', 7] float rows [s, action, o, t1, t2, x_end, y_end]: the window's start / end in units of each
    action's duration, actions entirely outside dropped, and the subject's final position for _slide / _pick_place
    (data/cater.py:446-466)."""
    f1, f2 = actions[:, 3].float(), actions[:, 4].float()
    t1, t2 = (start - f1) / (f2 - f1 + 1), (end - f1) / (f2 - f1 + 1)
    rows = torch.cat([actions[:, :3].float(), torch.stack([t1, t2], dim=-1)], dim=-1)
    inside = ~((t1 > 1) | (t2 < 0))
    rows = rows[inside]
    final = boxes_all[f2[inside].long(), rows[:, 0].long()][:, :2].clone()
    moving = (rows[:, 1] == vocab['action_name_to_idx']['_pick_place']) | (rows[:, 1] == vocab['action_name_to_idx']['_slide'])
    final[~moving] = 0.0
    return torch.cat([rows, final], dim=1)


def load_frames(paths, image_size):
    """PNG frames -> [F, 3, H, W] float32, bilinear resize, ImageNet mean / std (data/cater.py:137-149, group_transforms.py:14-20)."""
    from PIL import Image
    H, W = image_size
    out = []
    for p in paths:
        im = Image.open(p).convert('RGB').resize((W, H), Image.BILINEAR)
        a = np.asarray(im, np.float32).transpose(2, 0, 1) / 255.0
        out.append((a - np.array(IMG_MEAN, np.float32)[:, None, None]) / np.array(IMG_STD, np.float32)[:, None, None])
    return torch.from_numpy(np.stack(out))


class CATERScenes(Dataset):
    """Counterpart of the reference's CATERDataset (data/cater.py:17-466) over ``data_root/{scenes/*.json, videos/<id>/*.png}``.
    ``list_file``: lines "<video>.avi <labels>" selecting the split (data/cater.py:76-88); None = every scene file."""

    def __init__(self, data_root, vocab, list_file=None, is_test=False, nframes=301, image_size=(64, 64), frames_per_action=16,
                 initial_frames_per_sample=48, max_samples=None):
        self.root, self.vocab, self.is_test, self.nframes = data_root, vocab, is_test, nframes
        self.image_size, self.frames_per_action, self.span = tuple(image_size), frames_per_action, initial_frames_per_sample
        self.max_samples = max_samples
        names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(data_root, 'scenes', '*.json')))
        if list_file is not None:
            wanted = [line.split(' ')[0].split('.')[0] for line in open(list_file) if line.strip()]
            names = [n for n in wanted if n in set(names)]
        self.names = names
        self.scenes = {n: json.load(open(os.path.join(data_root, 'scenes', n + '.json'))) for n in names}

    def __len__(self):
        return len(self.names) if self.max_samples is None else min(len(self.names), self.max_samples)

    def frame_paths(self, name):
        paths = sorted(glob.glob(os.path.join(self.root, 'videos', name, '*.png')))
        return paths if len(paths) == self.nframes else None

    def __getitem__(self, index):
        name = self.names[index]
        sg = self.scenes[name]
        paths = self.frame_paths(name)
        if paths is None:                                     # the reference returns an all-None item; collate_fn drops it
            return None, None, None, None, None, None
        actions, (start, end) = choose_window(scene_actions(sg, self.vocab), len(paths) - 1, self.span, self.is_test)
        frames = list(range(start, end))[0:self.span:self.span // self.frames_per_action]
        try:
            vids = load_frames([paths[f] for f in frames], self.image_size)
        except Exception:
            print("Error: Failed to load frames in video id: {}".format(name))
            return None, None, None, None, None, None
        boxes_all = scene_boxes(sg)
        boxes = boxes_all[frames]
        return (vids, scene_objects(sg, self.vocab), boxes, frame_triplets(len(frames), boxes.size(1) - 1, self.vocab),
                normalize_actions(actions, boxes_all, start, end, self.vocab), '%s_%d-%d' % (name, start, end))


def collate_fn(vocab, batch):
    """Pad a list of dataset items into (imgs [B,F,3,H,W], objs [B,O,A] i64, boxes [B,F,O,4], triplets [B,F,T,3] i64,
    actions [B,Aact,7], ids) exactly as data/dataset_params.py:8-104: objects padded with 0, boxes with -1, triplets with
    [0, __padding__, 0], actions with [0, __padding__, 0, 0, 0, 0, 0]; failed items (first field None / bool) are dropped; an empty or
    inconsistent batch yields six Nones."""
    batch = [it for it in batch if it[0] is not None and not isinstance(it[0], bool)]
    if not batch:
        return None, None, None, None, None, None
    O = max(next(iter(it[1].values())).size(0) for it in batch)
    T = max(it[3].size(1) for it in batch)
    A = max(it[4].size(0) for it in batch)
    pad_t = torch.tensor([0, vocab['pred_name_to_idx']['__padding__'], 0], dtype=torch.long)
    pad_a = torch.tensor([0, vocab['action_name_to_idx']['__padding__'], 0, 0, 0, 0, 0], dtype=torch.float32)
    vids, objs_b, boxes_b, trip_b, acts_b, ids = [], [], [], [], [], []
    for vid, objs, boxes, triplets, actions, vid_id in batch:
        n, F = boxes.size(1), boxes.size(0)
        table = torch.zeros(O, len(objs), dtype=torch.long)
        for col, attr in enumerate(objs):                     # attribute order = the dict's own order, as in the reference
            table[:n, col] = objs[attr]
        objs_b.append(table)
        boxes_b.append(torch.cat([boxes, boxes.new_full((F, O - n, 4), -1.0)], dim=1))
        trip_b.append(torch.cat([triplets, pad_t.expand(F, T - triplets.size(1), 3)], dim=1))
        acts_b.append(torch.cat([actions.float(), pad_a.expand(A - actions.size(0), 7)], dim=0))
        vids.append(vid)
        ids.append(vid_id)
    try:
        return torch.stack(vids), torch.stack(objs_b), torch.stack(boxes_b), torch.stack(trip_b), torch.stack(acts_b), ids
    except Exception as e:
        print("Exception in {}".format(e))
        return None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ synthetic scene files
_ACTIONS = ('_slide', '_pick_place', '_rotate', '_contain', '_no_op')


def synthetic_scene(seed, nframes=301, n_objects=None):
    """A scene dictionary with the schema of CATER's scene JSON: objects (instance, shape, color, material, size, per-frame 3-D
    ``locations``) that sit still or glide between waypoints, and the matching ``movements`` table."""
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.randint(3, 8)) if n_objects is None else n_objects
    shapes, colors = ('cube', 'sphere', 'cylinder', 'spl', 'cone'), ('gray', 'red', 'blue', 'green', 'brown', 'purple', 'cyan', 'yellow', 'gold')
    objects, movements = [], {}
    for i in range(n):
        inst = 'obj_%d' % i
        pos = rs.uniform(-2.5, 2.5, size=2)
        track, moves, f = np.zeros((nframes, 3)), [], 0
        track[:, 2] = rs.uniform(0.3, 0.6)
        while f < nframes - 1:
            dur = int(rs.randint(8, 60))
            g = min(nframes - 1, f + dur)
            action = _ACTIONS[int(rs.randint(0, len(_ACTIONS)))]
            target = rs.uniform(-2.5, 2.5, size=2) if action in ('_slide', '_pick_place') else pos
            steps = np.linspace(0.0, 1.0, g - f + 1)[:, None]
            track[f:g + 1, :2] = pos[None] * (1 - steps) + target[None] * steps
            other = 'obj_%d' % int(rs.randint(0, n)) if action == '_contain' else None
            if action != '_no_op' or rs.rand() < 0.5:
                moves.append([action, other, int(f), int(g)])
            pos, f = target, g + int(rs.randint(0, 10))
            track[g:, :2] = pos[None]
        objects.append({'instance': inst, 'shape': shapes[int(rs.randint(0, 5))], 'color': colors[int(rs.randint(0, 9))],
                        'material': ('rubber', 'metal')[int(rs.randint(0, 2))], 'size': ('small', 'large', 'medium')[int(rs.randint(0, 3))],
                        'locations': {str(k): [float(v) for v in track[k]] for k in range(nframes)}})
        movements[inst] = moves
    if not any(m[3] - m[2] >= MIN_ACTION_FRAMES for ms in movements.values() for m in ms):
        movements['obj_0'].append(['_rotate', None, 5, 5 + 2 * MIN_ACTION_FRAMES])
    return {'objects': objects, 'movements': movements}


def write_synthetic_dataset(root, n_videos=4, nframes=301, seed=0, frame_size=(FRAME_H, FRAME_W)):
    """scenes/<name>.json + videos/<name>/%05d.png (seeded noise frames) + lists/train.txt under ``root``; returns the list file."""
    from PIL import Image
    os.makedirs(os.path.join(root, 'scenes'), exist_ok=True)
    os.makedirs(os.path.join(root, 'lists'), exist_ok=True)
    names = []
    for v in range(n_videos):
        name = 'CATER_syn_%06d' % (seed * 1000 + v)
        names.append(name)
        json.dump(synthetic_scene(seed * 1000 + v, nframes), open(os.path.join(root, 'scenes', name + '.json'), 'w'))
        d = os.path.join(root, 'videos', name)
        os.makedirs(d, exist_ok=True)
        rs = np.random.RandomState(seed * 1000 + v)
        base = rs.randint(0, 255, size=frame_size + (3,)).astype(np.uint8)
        for k in range(nframes):
            Image.fromarray(np.roll(base, k, axis=1)).save(os.path.join(d, '%05d.png' % k))
    lst = os.path.join(root, 'lists', 'train.txt')
    with open(lst, 'w') as f:
        for n in names:
            f.write('%s.avi 0\n' % n)
    return lst


def build_loaders(args, vocab, nframes=301):
    """(train_loader, train_graph_loader) over ``args.data_root`` with the clip shapes of the reference's configuration table
    (data/dataset_params.py:158-171): short clips of frames_per_action frames drawn from a window 3x as long for the G/D step, and
    4 * frames_per_action_graph frames (window 3x) for the graph step; batch_size, shuffle, drop_last as scripts/train.py:116-134."""
    from functools import partial
    from torch.utils.data import DataLoader
    lst = getattr(args, 'train_list', None) or os.path.join(args.data_root, 'lists', 'train.txt')
    lst = lst if os.path.exists(lst) else None
    fpa, fpg = args.frames_per_action, 4 * args.frames_per_action_graph
    mk = lambda f: CATERScenes(args.data_root, vocab, list_file=lst, nframes=nframes, image_size=args.image_size, frames_per_action=f,
                               initial_frames_per_sample=3 * f)
    kw = dict(batch_size=args.batch_size, shuffle=True, drop_last=True, num_workers=args.loader_num_workers,
              collate_fn=partial(collate_fn, vocab))
    return DataLoader(mk(fpa), **kw), DataLoader(mk(fpg), **kw)
