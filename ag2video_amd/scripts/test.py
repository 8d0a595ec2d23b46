"""Inference counterpart of the reference's scripts/test.py (main: :170-301; scripted action edits: :47-148) on seeded synthetic
clips: restore a checkpoint written by scripts.train (+ its run_args.json), put the model in eval mode and generate every clip
autoregressively (``test_mode=True``) -- once from its own action graph (``--save_test``: ground-truth or predicted layout,
``--use_test``) and once per scripted edit of the action graph with the model's predicted boxes (``--save_actions``).

    python -m ag2video_amd.scripts.test --checkpoint out/latest.pt --gpu_ids 0 --output_dir results

The reference writes .gif / .pkl through cv2 / imageio (not installed here); this driver stores the raw tensors as .npz.
"""
import argparse
import json
import os
from argparse import Namespace

import numpy as np
import torch

from ..data.args import init_args
from ..data.synthetic import make_clip_batch
from ..data.vocab import cater_vocab, smth_vocab
from ..models.meta_models import AG2VideoModel
from ..models.utils import batch_to

# Rows [subject, action id, object, t1, t2, x_end, y_end] as in the collated ``actions`` tensor.  Something-Else edits move object 1
# (actions 1..4 = up / right / down / left in its vocabulary: scripts/test.py:47-96).
_SMTH_EDITS = {
    'up': [[1, 1, 0, 0, 1.01, 0, 0]],
    'right': [[1, 2, 0, 0, 1.01, 0, 0]],
    'down': [[1, 3, 0, 0, 1.01, 0, 0]],
    'left': [[1, 4, 0, 0, 1.01, 0, 0]],
    'right_up': [[1, 1, 0, 0, 1.01, 0, 0], [1, 2, 0, 0, 1.01, 0, 0]],
    'down_left': [[1, 3, 0, 0, 1.01, 0, 0], [1, 4, 0, 0, 1.01, 0, 0]],
}


def _converge(num_objs, start_of, end):
    """All objects but the first start action 3 towards object 0, object j at relative time start_of(j / num_objs)
    (scripts/test.py:99-133: converge / converge_before / converge_after differ in the start offsets and the end time)."""
    return [[j, 3, 0, start_of(float(j) / num_objs), end, 0, 0] for j in range(1, num_objs)]


def _swap(boxes, rng):
    """Two random objects trade places: _pick_place / _slide to the other's first-frame position (scripts/test.py:136-147)."""
    n = boxes.shape[2] - 1
    i, j = rng.permutation(n)[:2]
    ti, tj = boxes[0, 0, j], boxes[0, 0, i]
    return [[i, 5, i, -0.3, 1.3, float(ti[0]), float(ti[1])], [j, 2, j, -0.3, 1.3, float(tj[0]), float(tj[1])]]


def scripted_edits(dataset, objs, boxes, rng, actions=None, vocab=None):
    """-> {edit name: {'now' | 'before' | 'after': actions tensor [1, A, 7]}}: the edit itself and its two re-timed variants
    (scripts/test.py:228-262: t1/t2 = (0, 3) "before", (-2, 1) "after"; converge has its own variants).
    * Something-Else: ``down_left`` / ``right_up`` are skipped when the first object of the first frame sits where the move would
      leave the image (x < 0.35 or y > 0.65, resp. x > 0.65 or y < 0.35: scripts/test.py:216-221).
    * ``actions`` (the clip's own collated action rows) adds the reference's per-clip "initial action" edit (scripts/test.py:198-214):
      one randomly chosen action of the clip, named by the vocabulary, re-timed to (0, 1.1) -- CATER keeps only that row,
      Something-Else re-times all rows."""
    def T(rows):
        return torch.tensor(rows, dtype=torch.float32).unsqueeze(0)

    def retimed(rows, t1, t2):
        return [[r[0], r[1], r[2], t1, t2, r[5], r[6]] for r in rows]

    def triple(rows):
        return {'now': T(rows), 'before': T(retimed(rows, 0.0, 3.0)), 'after': T(retimed(rows, -2.0, 1.0))}
    out = {}
    if dataset == 'smth_else':
        first = boxes[0, 0, 0].detach().cpu().numpy()
        for name, rows in _SMTH_EDITS.items():
            if name == 'down_left' and (first[0] < 0.35 or first[1] > 0.65):
                continue
            if name == 'right_up' and (first[0] > 0.65 or first[1] < 0.35):
                continue
            out[name] = triple(rows)
    else:
        n = objs.shape[1] - 1
        if n >= 2:
            out['converge'] = {'now': T(_converge(n, lambda s: s, 1.05)), 'before': T(_converge(n, lambda s: -s, 2.0)),
                               'after': T(_converge(n, lambda s: -s - 1.0, 1.05))}
            out['swap'] = triple(_swap(boxes.cpu().numpy(), rng))
    if actions is not None and actions.numel() and vocab is not None:
        rows = actions[0].detach().cpu().numpy().tolist()
        pick = int(rng.permutation(len(rows))[0])
        aid = int(rows[pick][1])
        names = vocab['action_idx_to_name']
        # (Something-Else: a list indexed by the action id; a CATER vocabulary read from JSON has string keys: test.py:203-205)
        name = names[aid] if isinstance(names, (list, tuple)) or aid in names else names[str(aid)]
        chosen = retimed(rows[pick:pick + 1] if dataset != 'smth_else' else rows, 0.0, 1.1)
        out.setdefault(str(name), triple(chosen))
    return out


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--checkpoint', type=str, required=True)
    p.add_argument('--gpu_ids', type=int, default=0)
    p.add_argument('--save_test', type=int, default=1)
    p.add_argument('--save_actions', type=int, default=1)
    p.add_argument('--output_dir', type=str, required=True)
    p.add_argument('--use_test', type=int, default=1)
    # build-specific: the synthetic validation clips
    p.add_argument('--num_clips', type=int, default=4)
    p.add_argument('--frames', type=int, default=None)
    p.add_argument('--seed', type=int, default=4321)
    p.add_argument('--trust_checkpoint', type=int, default=0,
                   help='1: unpickle arbitrary objects from the checkpoint (torch.load(weights_only=False), as the reference does). '
                        'Checkpoints written by scripts.train hold tensors and plain containers only and load without it.')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    run_dir = os.path.dirname(os.path.abspath(args.checkpoint))
    train_args = json.load(open(os.path.join(run_dir, 'run_args.json')))
    train_args.update(gpu_ids=[args.gpu_ids], batch_size=1, loader_num_workers=0, frames_per_action_graph=1)   # test.py:174-178
    opt = Namespace(**train_args)
    opt.image_size = tuple(opt.image_size)
    if not torch.cuda.is_available():
        raise RuntimeError("ag2video_amd.scripts.test needs an MI355X: there is no CPU fallback")
    device = torch.device('cuda', args.gpu_ids)
    torch.cuda.set_device(device)
    ckpt = torch.load(args.checkpoint, map_location=device, weights_only=not args.trust_checkpoint)
    # the vocabulary the model was trained with travels in the checkpoint (scripts/train.py:528-543); the defaults only stand in for
    # a checkpoint without one
    opt.vocab = ckpt.get('vocab') or (smth_vocab() if opt.dataset == 'smth_else' else cater_vocab())
    init_args(opt)
    model = AG2VideoModel(opt, device)
    model.load_state_dict(ckpt['model_state'], strict=True)
    model.eval()
    out_root = os.path.join(args.output_dir, 'results_%s' % opt.dataset, os.path.basename(run_dir) + ('_gt_layout' if args.use_test else ''))
    os.makedirs(out_root, exist_ok=True)
    frames = args.frames or opt.frames_per_action
    lo, hi = (2, 3) if opt.dataset == 'smth_else' else tuple(getattr(opt, 'synthetic_objects', (5, 10)))
    rng = np.random.RandomState(args.seed)
    n_edits = 0

    def run(vids, objs, triplets, actions, boxes, use_gt):
        with torch.no_grad():
            return model(vids, objs, triplets, actions.to(device), boxes_gt=boxes, test_mode=True, use_gt=use_gt)

    def dump(path, out, vids):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, imgs_pred=out[0][0].cpu().numpy(), boxes_pred=out[1][0].cpu().numpy(), imgs_gt=vids[0].cpu().numpy())
    for c in range(args.num_clips):
        clip = make_clip_batch(opt.vocab, 1, frames, opt.image_size, seed=args.seed + c, min_objs=lo, max_objs=hi)
        vids, objs, boxes, triplets, actions, vid_id = batch_to(clip, device)
        if args.save_actions:
            for name, variants in scripted_edits(opt.dataset, objs, boxes, rng, actions, opt.vocab).items():
                for when, acts in variants.items():
                    sub = 'action_accuracy/%s' % name if when == 'now' else 'action_timing/%s/%s' % (name, when)
                    dump(os.path.join(out_root, sub, vid_id[0] + '.npz'), run(vids, objs, triplets, acts, boxes, False), vids)
                    n_edits += 1
                # the ground-truth clip beside every edit (scripts/test.py:259-261)
                gt = os.path.join(out_root, 'gt_action', name, vid_id[0] + '.npz')
                os.makedirs(os.path.dirname(gt), exist_ok=True)
                np.savez(gt, imgs_gt=vids[0].cpu().numpy())
        if args.save_test:
            dump(os.path.join(out_root, 'test', vid_id[0] + '.npz'), run(vids, objs, triplets, actions, boxes, bool(args.use_test)), vids)
    return {'clips': args.num_clips, 'edits': n_edits, 'output': out_root}


if __name__ == '__main__':
    print(main())
