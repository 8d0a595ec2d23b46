"""Pins of the SURVEY 8(f) "next" rows and D4 against golden vectors of the unmodified reference classes
(tests/golden/extras.npz, generator: tests/golden/make_golden_extras.py): the oracle's VGG perceptual loss and temporal
discriminator; the product's host-side data path (collate_fn, CATER scene -> tensors) which is plain integer / small float work
and therefore compared (bit-)exactly on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(HERE, 'golden', 'extras.npz'), allow_pickle=False)


def test_oracle_vgg_loss_matches_reference(gold):
    from oracle import model as M
    shapes = json.loads(str(gold['vgg.keys']))
    assert len(shapes) == 26 and 'vgg.slice4.19.weight' in shapes and shapes['vgg.slice5.28.weight'] == [512, 512, 3, 3]
    V = {k: v.detach() for k, v in M.make_state(shapes, 3, 'V.').items()}
    x = torch.from_numpy(gold['vgg.x']).requires_grad_(True)
    y = torch.from_numpy(gold['vgg.y'])
    loss = M.vgg_loss(V, x, y)
    loss.backward()
    assert abs(loss.item() - float(gold['vgg.loss'])) <= 1e-5 * abs(float(gold['vgg.loss']))
    ref = gold['vgg.grad_x']
    assert np.abs(x.grad.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    feats = M.vgg19_slices(V, x.detach().view(-1, 3, 32, 32))
    for i, f in enumerate(feats):
        st = gold['vgg.feat%d_stats' % i]
        assert tuple(f.shape[1:3]) == (int(st[2]), int(st[3])) and abs(f.abs().sum().item() - st[1]) <= 1e-4 * st[1]


@pytest.mark.parametrize("tag,side,ndf,seed", [('dtemp', 32, 8, 5), ('dtemp64', 64, 16, 7)])
def test_oracle_temporal_discriminator_matches_reference(gold, tag, side, ndf, seed):
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    from oracle import model as M
    opt = make_opt(cater_vocab(), ['--gpu_ids', '-1', '--no_vgg_loss', '--image_size', '%d,%d' % (side, side), '--batch_size', '2',
                                  '--ndf', str(ndf), '--frames_per_action', '4'])
    shapes = json.loads(str(gold[tag + '.keys']))
    PT = M.make_state(shapes, seed, 'T.')
    img = torch.from_numpy(gold[tag + '.img']).requires_grad_(True)
    res = M.temporal_discriminator_forward(PT, opt, img, training=True)
    tot = 0
    for i, scale in enumerate(res):
        for j, f in enumerate(scale):
            ref = gold[tag + '.out%d_%d' % (i, j)]
            assert np.abs(f.detach().numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), (i, j)
            tot = tot + (f * torch.linspace(-1, 1, f.numel()).view_as(f)).sum()
    tot.backward()
    ref = gold[tag + '.grad_img']
    assert np.abs(img.grad.numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    for k, t in PT.items():
        if k.endswith('weight_u'):
            assert np.abs(t.numpy() - gold[tag + '.state.' + k]).max() <= 1e-5
        if t.grad is not None and (tag + '.grad.' + k) in gold.files:
            got = np.array([t.grad.double().abs().sum().item(), t.grad.double().pow(2).sum().sqrt().item()])
            assert np.allclose(got, gold[tag + '.grad.' + k], rtol=2e-3, atol=1e-6), k


def test_collate_fn_matches_reference(gold):
    from ag2video_amd.data.cater import collate_fn
    from ag2video_amd.data.vocab import cater_vocab
    vocab = cater_vocab()
    items = []
    for i in range(4):
        if ('collate.in%d.vid' % i) not in gold.files:
            items.append((False, None, None, None, None, None))
            continue
        objs = {a: torch.from_numpy(gold['collate.in%d.objs.%s' % (i, a)]) for a in vocab['attributes']}
        items.append((torch.from_numpy(gold['collate.in%d.vid' % i]), objs, torch.from_numpy(gold['collate.in%d.boxes' % i]),
                      torch.from_numpy(gold['collate.in%d.triplets' % i]), torch.from_numpy(gold['collate.in%d.actions' % i]),
                      'vid%d' % (i if i == 0 else i - 1)))
    out = collate_fn(vocab, items)
    for n, t in zip(('vids', 'objs', 'boxes', 'triplets', 'actions'), out[:5]):
        ref = gold['collate.out.' + n]
        assert t.numpy().dtype == ref.dtype and np.array_equal(t.numpy(), ref), n
    assert out[5] == json.loads(str(gold['collate.out.ids']))
    assert collate_fn(vocab, [(None,) * 6]) == (None,) * 6             # nothing usable: six Nones (dataset_params.py:102-104)


@pytest.mark.parametrize("seed", [0, 1])
def test_cater_scene_tensors_match_reference(gold, seed):
    from ag2video_amd.data import cater as C
    from ag2video_amd.data.vocab import cater_vocab
    vocab = cater_vocab()
    sg = C.synthetic_scene(seed, nframes=301)
    p = 'cater.s%d.' % seed
    for a, v in C.scene_objects(sg, vocab).items():
        assert np.array_equal(v.numpy(), gold[p + 'objs.' + a]), a
    boxes = C.scene_boxes(sg)
    assert boxes.dtype == torch.float32 and np.abs(boxes.numpy() - gold[p + 'boxes_all']).max() <= 1e-6
    rows = C.scene_actions(sg, vocab)
    actions, (s, e) = C.choose_window(rows, 300, 48, True)
    assert (s, e) == tuple(gold[p + 'window']) and np.array_equal(actions.numpy(), gold[p + 'actions_raw'])
    norm = C.normalize_actions(actions, boxes, s, e, vocab)
    assert np.abs(norm.numpy() - gold[p + 'actions_norm']).max() <= 1e-6
    frames = list(range(s, e))[0:48:48 // 16]
    assert frames == list(gold[p + 'frames_lst'])
    assert np.array_equal(C.frame_triplets(len(frames), boxes.size(1) - 1, vocab).numpy(), gold[p + 'triplets'])
    np.random.seed(11 + seed)
    actions_t, (s2, e2) = C.choose_window(rows, 300, 48, False)
    assert (s2, e2) == tuple(gold[p + 'train_window'])
    assert np.abs(C.normalize_actions(actions_t, boxes, s2, e2, vocab).numpy() - gold[p + 'train_actions_norm']).max() <= 1e-6


def test_cater_loader_end_to_end(tmp_path):
    """Scene files + cached frames on disk -> DataLoader with collate_fn -> a batch with the training step's contract."""
    from functools import partial
    from ag2video_amd.data import cater as C
    from ag2video_amd.data.vocab import cater_vocab
    vocab = cater_vocab()
    root = str(tmp_path)
    lst = C.write_synthetic_dataset(root, n_videos=3, nframes=70, seed=2, frame_size=(24, 32))
    ds = C.CATERScenes(root, vocab, list_file=lst, is_test=True, nframes=70, image_size=(16, 16), frames_per_action=4,
                       initial_frames_per_sample=48)
    assert len(ds) == 3
    vdir = os.path.join(root, 'videos', ds.names[1])
    os.remove(os.path.join(vdir, sorted(os.listdir(vdir))[0]))                     # a video whose frame cache is incomplete
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=partial(C.collate_fn, vocab))
    imgs, objs, boxes, triplets, actions, ids = next(iter(loader))
    assert imgs.shape == (2, 4, 3, 16, 16) and imgs.dtype == torch.float32          # the video with a missing frame is dropped
    B, O = objs.shape[0], objs.shape[1]
    assert objs.shape == (2, O, 4) and objs.dtype == torch.int64 and boxes.shape == (2, 4, O, 4)
    assert triplets.shape[:2] == (2, 4) and triplets.shape[3] == 3 and actions.shape[2] == 7 and len(ids) == 2
    assert (objs[:, -1] == 0).all() or O > 0
    for b in range(B):
        n = int((objs[b].sum(1) != 0).sum())                                       # real objects; then the __image__ row, then padding
        assert torch.equal(boxes[b, :, n], torch.tensor([0.0, 0.0, 1.0, 1.0]).expand(4, 4))
        assert (boxes[b, :, n + 1:] == -1).all()


def test_perceptual_loss_needs_weights_or_an_explicit_opt_in():
    """the default flags (no --no_vgg_loss, lambda_vgg 10) must not silently optimise a randomly initialised VGG19."""
    import pytest
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.models.spade_models.loss_model import LossModel
    flags = ['--image_size', '32,32', '--gpu_ids', '-1']
    with pytest.raises(RuntimeError, match='vgg_weights'):
        LossModel(make_opt(cater_vocab(), flags), discriminator=None)
    lm = LossModel(make_opt(cater_vocab(), flags + ['--vgg_random_init']), discriminator=None)
    assert hasattr(lm, 'criterionVGG') and lm.vgg_weights_loaded is False
    assert not hasattr(LossModel(make_opt(cater_vocab(), flags + ['--no_vgg_loss']), discriminator=None), 'criterionVGG')


def test_something_else_reader_matches_reference(tmp_path):
    """ag2video_amd.data.smth against the reference's SmthElseDataset on the same miniature dataset (tests/golden/smth.npz, generator
    tests/golden/make_golden_smth.py): vocabulary, label filters, unique frames, test / train windows, box tracks, hand-chain
    actions, spatial triplets (Python ``random``) and whole items, bit for bit; frames by the CATER reader's loader; a batch through
    collate_fn."""
    import pickle
    import random
    import sys
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    from make_golden_smth import SPAN, build_fixture
    from ag2video_amd.data import smth as S
    from ag2video_amd.data.cater import collate_fn, load_frames
    gold = np.load(os.path.join(HERE, 'golden', 'smth.npz'), allow_pickle=False)
    root, labels = build_fixture(str(tmp_path))
    ds = S.SmthElseScenes(root, labels, image_size=(32, 32), frames_per_action=SPAN, initial_frames_per_sample=SPAN)
    ref_vocab = json.loads(str(gold['vocab']))
    mine = json.loads(json.dumps({k: ds.vocab[k] for k in ref_vocab}, sort_keys=True))
    assert mine == ref_vocab
    assert [r['id'] for r in ds.rows] == gold['kept_ids'].tolist() and len(ds) == 6
    for k, row in enumerate(ds.rows):
        p = 'v%d.' % k
        with open(os.path.join(root, 'offical_release_boxes', 'boxes_by_video_id', str(row['id']), 'boxes.pkl'), 'rb') as f:
            track = S.unique_frames(pickle.load(f))
        assert [e['name'] for e in track] == json.loads(str(gold[p + 'unique_names']))
        n_inst = S.INSTANCES_PER_TEMPLATE[row['template']]
        assert np.array_equal(np.array(S.action_window(track, n_inst, SPAN, True), np.float64), gold[p + 'window_test'])
        np.random.seed(100 + k)
        f1, f2, s, e = S.action_window(track, n_inst, SPAN, False)
        assert np.array_equal(np.array([f1, f2, s, e], np.float64), gold[p + 'window_train'])
        boxes, objs = S.track_boxes(track[f1:f2], (120, 90), n_inst, ds.objs_mapping, ds.vocab)
        assert np.array_equal(boxes.numpy(), gold[p + 'boxes']) and np.array_equal(objs['object'].numpy(), gold[p + 'objs'])
        acts = S.hand_chain_actions(objs['object'], ds.vocab, ds.vocab['action_name_to_idx'][row['template']], s, e)
        assert np.array_equal(acts.numpy(), gold[p + 'actions'])
        random.seed(200 + k)
        assert np.array_equal(S.spatial_triplets(boxes, ds.vocab).numpy(), gold[p + 'triplets'])
        for mode, is_test in (('train', False), ('test', True)):
            ds.is_test = ds.is_val = is_test
            np.random.seed(300 + k)
            random.seed(400 + k)
            vids, o, b, t, a, vid_id = ds[k]
            assert vid_id == str(gold[p + mode + '.id'])
            assert np.array_equal(o['object'].numpy(), gold[p + mode + '.objs']) and np.array_equal(b.numpy(), gold[p + mode + '.boxes'])
            assert np.array_equal(t.numpy(), gold[p + mode + '.triplets']) and np.array_equal(a.numpy(), gold[p + mode + '.actions'])
            assert tuple(vids.shape) == (int(gold[p + mode + '.n_frames']), 3, 32, 32) and vids.dtype == torch.float32
        ds.is_test = ds.is_val = False
    # the frames of an item are the window's frames through the CATER reader's loader (PIL bilinear resize, ImageNet mean / std)
    np.random.seed(300)
    random.seed(400)
    vids, _, _, _, _, vid_id = ds[0]
    f1 = int(vid_id.split('_')[1].split('-')[0])
    paths = sorted(os.path.join(root, 'frames', e['name']) for e in ds.track(ds.rows[0]['id']))
    assert torch.equal(vids, load_frames(paths[f1:f1 + SPAN], (32, 32)))
    # a video whose hand never appears is skipped, and the batch still collates (dataset_params.py:8-104)
    bad = ds.track(ds.rows[1]['id'])
    for e in bad:
        e['labels'] = [lab for lab in e['labels'] if lab['category'] != 'hand']
        e['nr_instances'] = len(e['labels'])
    with open(os.path.join(root, 'offical_release_boxes', 'boxes_by_video_id', str(ds.rows[1]['id']), 'boxes.pkl'), 'wb') as f:
        pickle.dump(bad, f)
    assert ds[1][0] is None
    batch = collate_fn(ds.vocab, [ds[i] for i in range(len(ds))])
    assert tuple(batch[0].shape) == (5, SPAN, 3, 32, 32) and batch[1].shape[0] == 5 and batch[1].shape[2] == 1
    assert batch[2].shape[2] == 3 and (batch[2][:, :, -1] == -1).any()        # two-instance videos padded to the covering video's three
