"""Golden vectors for the Something-Else reader (runs ONLY in the build container, where /root/reference exists).

Imports the *unmodified* reference SmthElseDataset (data/smth.py; shims as in make_golden.py: torchvision's transform classes are
no-ops there, so frame PIXELS are not part of these vectors -- they are produced by the same PIL resize + normalisation as the CATER
reader's, which the CATER fixtures pin) on the miniature dataset ``build_fixture`` writes, and stores data only ->
tests/golden/smth.npz:

  vocab / kept_ids        the vocabulary and the label rows SmthElseDataset.__init__ keeps (data/smth.py:79-115)
  v<k>.*                  per video: unique frames, the test / train windows, boxes + object ids, the hand-chain actions, the
                          spatial triplets (Python ``random`` seeded), and the full __getitem__ tuple without the frames
``build_fixture`` does not touch the reference: tests/test_extras_cpu.py calls it to recreate the same dataset files.
"""
import csv
import json
import os
import pickle
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

N_VIDEOS, N_FRAMES, SPAN = 6, 24, 8      # six templates: the last one ('Covering ...') tracks three instances


def build_fixture(root):
    """The synthetic dataset of ag2video_amd.data.smth.write_synthetic_dataset plus label rows every filter of the reader has to
    drop, one video with repeated frame names and one with a box missing in one frame.  -> (data_root, labels.csv)"""
    from ag2video_amd.data.smth import ACTION_TEMPLATES, synthetic_track, write_synthetic_dataset
    root, labels = write_synthetic_dataset(root, n_videos=N_VIDEOS, n_frames=N_FRAMES, seed=0)
    rows = list(csv.reader(open(labels, newline='')))
    t0 = ACTION_TEMPLATES[0]
    rows += [['90', '9001', 'x', t0, "['pen']", '2', 'False'],                       # a recorded error (any non-empty cell)
             ['91', '9002', 'x', 'Folding [something]', "['pen']", '2', ''],          # template outside the vocabulary
             ['92', '9003', 'x', t0, "['pen']", '3', ''],                             # instance count of another template
             ['93', '9004', 'x', t0, "['a unicorn']", '2', '']]                       # placeholder unknown to the object mapping
    with open(labels, 'w', newline='') as f:
        csv.writer(f).writerows(rows)
    # video 1001: every frame listed twice; video 1002: one object box missing in one frame
    for vid, kw, dup in ((1001, {}, True), (1002, {'hide': (0, 9)}, False)):
        template = ACTION_TEMPLATES[(vid - 1000) % (len(ACTION_TEMPLATES) - 1)]
        track, _ = synthetic_track(vid, template, N_FRAMES, (120, 90), **kw)
        if dup:
            track = [e for e in track for _ in (0, 1)]
        with open(os.path.join(root, 'offical_release_boxes', 'boxes_by_video_id', str(vid), 'boxes.pkl'), 'wb') as f:
            pickle.dump(track, f)
    return root, labels


def main():
    from make_golden import install_shims
    install_shims()
    import types
    for name in ('cv2', 'skvideo', 'skvideo.io'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.argv = ['make_golden_smth']
    from data.smth import SmthElseDataset, clean_boxes_metadata
    from data.SomethingElse.config import action_to_num_objects
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root, labels = build_fixture(tmp)
        ds = SmthElseDataset(data_root=root, labels=labels, image_size=(32, 32), frames_per_action=SPAN, initial_frames_per_sample=SPAN)
        v = {k: ds.vocab[k] for k in ('action_idx_to_name', 'action_name_to_idx', 'pred_name_to_idx', 'attributes', 'reverse_attributes',
                                      'object_idx_to_name', 'object_name_to_idx')}
        out['vocab'] = np.array(json.dumps(v, sort_keys=True))
        out['kept_ids'] = np.array([int(i) for i in ds.vid_names])
        for k, vid in enumerate(ds.vid_names):
            p = 'v%d.' % k
            with open(os.path.join(root, 'offical_release_boxes', 'boxes_by_video_id', str(vid), 'boxes.pkl'), 'rb') as f:
                raw = pickle.load(f)
            track = clean_boxes_metadata(raw)
            out[p + 'unique_names'] = np.array(json.dumps([e['name'] for e in track]))
            template = ds.labels[ds.labels['id'] == vid].iloc[0]['template']
            n_inst = action_to_num_objects[template]
            f1, f2, s, e = ds.extract_actions_split(track, n_inst, True)
            out[p + 'window_test'] = np.array([f1, f2, s, e], np.float64)
            np.random.seed(100 + k)
            g1, g2, gs, ge = ds.extract_actions_split(track, n_inst, False)
            out[p + 'window_train'] = np.array([g1, g2, gs, ge], np.float64)
            sub = track[g1:g2]
            status, boxes, objs = ds.extract_bounding_boxes(sub, (120, 90), n_inst)
            assert status is True, boxes
            out[p + 'boxes'], out[p + 'objs'] = boxes.numpy(), objs['object'].numpy()
            st, acts = ds.extract_actions(objs, ds.vocab['action_name_to_idx'][template], gs, ge)
            assert st is True, acts
            out[p + 'actions'] = acts.numpy()
            random.seed(200 + k)
            out[p + 'triplets'] = ds.extract_triplets(boxes).numpy()
            # the whole item (frames excluded: the shimmed transforms return the PIL images)
            for mode, is_test in (('train', False), ('test', True)):
                ds.is_test, ds.is_val = is_test, is_test
                np.random.seed(300 + k)
                random.seed(400 + k)
                item = ds[k]
                assert item[0] is not False, item
                out[p + mode + '.objs'], out[p + mode + '.boxes'] = item[1]['object'].numpy(), item[2].numpy()
                out[p + mode + '.triplets'], out[p + mode + '.actions'] = item[3].numpy(), item[4].numpy()
                out[p + mode + '.id'] = np.array(item[5])
                out[p + mode + '.n_frames'] = np.array(len(item[0]))
            ds.is_test = ds.is_val = False
    np.savez_compressed(os.path.join(HERE, 'smth.npz'), **out)
    print('wrote smth.npz with %d arrays, %.1f KB' % (len(out), os.path.getsize(os.path.join(HERE, 'smth.npz')) / 1024))


if __name__ == '__main__':
    main()
