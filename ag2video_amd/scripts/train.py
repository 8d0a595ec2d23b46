"""Training loop counterpart of the reference's scripts/train.py (main: :335-513, checkpoints :27-57,528-543) on
seeded synthetic CATER-shaped clips (the real datasets need files/network that are out of scope).

    python -m ag2video_amd.scripts.train --batch_size 2 --image_size 64,64 --num_iterations 10 --no_vgg_loss
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m ag2video_amd.scripts.train ...     # one rank per GPU

Same flags, step order (G step -> D step -> graph step on a 4x longer clip), NaN guard (collective across ranks), per-step
try/except (single process only), loss-dict keys and checkpoint dictionary keys as the reference.
"""
import json
import logging
import os

import numpy as np
import torch

from .. import distributed as D
from ..data.args import get_args, init_args
from ..data.synthetic import make_clip_batch
from ..data.vocab import cater_vocab, smth_vocab
from ..engine import TrainStep


def save_checkpoint(args, path, ts, epoch, t):
    """Checkpoint dictionary with the reference's keys (scripts/train.py:528-543)."""
    ckpt = {
        'model_state': ts.model.state_dict(),
        'gans_model_state': ts.gans_model.state_dict(),
        'd_img_state': ts.discriminator.img_discriminator.state_dict(),
        'd_img_optim_state': ts.discriminator.optimizer_d_img.state_dict(),
        'optim_state_gen': ts.optimizer_generator.state_dict(),
        'optim_state_graph': ts.optimizer_graph.state_dict(),
        'vocab': args.vocab,
        'counters': {'t': t, 'epoch': epoch},
    }
    torch.save(ckpt, path)


def restore_checkpoint(args, ts, path):
    """scripts/train.py:27-57 (optimizer state is mapped to the device by load_state_dict)."""
    ckpt = torch.load(path, map_location=ts.device, weights_only=True)     # tensors and plain containers only (save_checkpoint)
    ts.model.load_state_dict(ckpt['model_state'])
    ts.gans_model.load_state_dict(ckpt['gans_model_state'])
    ts.discriminator.img_discriminator.load_state_dict(ckpt['d_img_state'])
    ts.discriminator.optimizer_d_img.load_state_dict(ckpt['d_img_optim_state'])
    ts.optimizer_generator.load_state_dict(ckpt['optim_state_gen'])
    if 'optim_state_graph' in ckpt:
        ts.optimizer_graph.load_state_dict(ckpt['optim_state_graph'])
    return ckpt['vocab'], ckpt['counters']['epoch'], ckpt['counters']['t']


def main(args):
    logger = logging.getLogger(__name__)
    rank, world, local = D.init_from_env()
    if args.dataset == 'smth_else' and args.data_root:
        from ..data.smth import dataset_vocab            # object names come from the dataset's own mapping file (data/smth.py:85-104)
        args.vocab = dataset_vocab(args.data_root)
    else:
        args.vocab = smth_vocab() if args.dataset == 'smth_else' else cater_vocab()
    init_args(args)
    if not torch.cuda.is_available():
        raise RuntimeError("ag2video_amd.scripts.train needs an MI355X: there is no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    args.gpu_ids = [local]
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        json.dump({k: v for k, v in vars(args).items() if k != 'vocab'}, open(os.path.join(args.output_dir, 'run_args.json'), 'w'))
    ts = TrainStep(args, device)
    epoch, t = 0, 0
    ckpt_path = os.path.join(args.output_dir, 'latest.pt')
    if args.restore_checkpoint and os.path.exists(ckpt_path):
        args.vocab, epoch, t = restore_checkpoint(args, ts, ckpt_path)
    lo, hi = args.synthetic_objects
    F_short, F_long = args.frames_per_action, 4 * args.frames_per_action_graph     # dataset_params.py:166-170
    G_losses, D_losses = {}, {}
    # --data_root: CATER scene files + cached frames through the reference's two loaders (scripts/train.py:116-134); otherwise
    # seeded synthetic clips of the same tensor contract.  Every rank draws its own shard (DataLoader shuffling is per process).
    short_it = long_it = None
    if args.data_root:
        torch.manual_seed(args.seed + rank)
        if args.dataset == 'smth_else':
            # Something-Else: label table + box tracks + frames (data/smth.py); the spatial relations and the training window are drawn
            # with Python's / numpy's global generators, as in the reference
            import random
            from ..data.smth import build_loaders as build_smth_loaders
            random.seed(args.seed + rank)
            np.random.seed((args.seed + rank) % (2 ** 32))
            short_loader, long_loader = build_smth_loaders(args, args.vocab)
        else:
            from ..data.cater import build_loaders
            short_loader, long_loader = build_loaders(args, args.vocab, nframes=args.data_nframes)

        def cycle(loader, name):
            while True:
                usable = 0
                for b in loader:
                    if b[0] is not None:          # collate_fn returns six Nones for an unusable batch (train.py:442)
                        usable += 1
                        yield b
                if usable == 0:
                    # every item unusable (e.g. the frame count of the scenes differs from --data_nframes, or fewer scenes than
                    # --batch_size with drop_last): the reference would spin forever; at world > 1 the other ranks would wait in a
                    # collective until the RCCL timeout
                    raise RuntimeError("%s loader over --data_root %s yielded no usable batch in a full pass (data_nframes=%s, batch_size=%s)"
                                       % (name, args.data_root, args.data_nframes, args.batch_size))
        short_it, long_it = cycle(short_loader, 'train'), cycle(long_loader, 'train_graph')

    def next_clip(kind, seed):
        if short_it is not None:
            return next(short_it if kind == 'short' else long_it)
        return make_clip_batch(args.vocab, args.batch_size, F_short if kind == 'short' else F_long, args.image_size, seed=seed,
                               min_objs=lo, max_objs=hi)
    while t < args.num_iterations:
        if t % args.checkpoint_every == 0 and rank == 0:
            save_checkpoint(args, os.path.join(args.output_dir, 'itr_%s.pt' % t), ts, epoch, t)
        seed = args.seed + 1000003 * t + rank
        # One code path with bench.py and the tests: TrainStep.gd_step / graph_step (the in-place all-reduce over the optimizers'
        # flat gradient buffers, the collective NaN guard).  Exceptions: the reference is one process and logs + continues
        # (train.py:466-468,495-497); with one process per GPU a rank that swallowed an exception would leave the others inside a
        # collective, so at world > 1 the error is logged, the process group torn down and the exception re-raised (torchrun then
        # stops the job with a non-zero exit code).
        if not args.graph_only:
            try:
                batch = ts.to_device(next_clip('short', seed))
                _, G_losses, D_new = ts.gd_step(batch)
                if D_new is None:                       # NaN guard tripped on some rank: train.py:450-453 (t += 1; continue)
                    t += 1
                    continue
                D_losses = D_new
            except Exception as e:                                                     # train.py:466-468
                print("Error in video_id: {}".format(seed))
                logger.exception(e)
                if D.is_dist():
                    D.shutdown()
                    raise
        G_graph_losses = {}
        try:
            gbatch = ts.to_device(next_clip('long', seed + 7))
            _, G_graph_losses = ts.graph_step(gbatch)
        except Exception as e:
            print("Error in video_id: {}".format(seed + 7))
            logger.exception(e)
            if D.is_dist():
                D.shutdown()
                raise
        if t % args.print_every == 0 and rank == 0:
            print('t = %d / %d' % (t, args.num_iterations))
            for name, val in G_graph_losses.items():
                print(' Graph [%s]: %.4f' % (name, val))
            for name, val in G_losses.items():
                print(' G [%s]: %.4f' % (name, val))
            for name, val in D_losses.items():
                print(' D [%s]: %.4f' % (name, val))
        t += 1
    ts.check_health()
    if rank == 0:
        save_checkpoint(args, ckpt_path, ts, epoch, t)


if __name__ == '__main__':
    main(get_args())
