"""Command-line flags of the AG2Video training step.

Same flag names and defaults as the reference parser (data/args.py:22-181) for
every flag the hot path or the training loop reads (SURVEY.md 8b lists them);
``init_args`` mirrors data/args.py:184-207 (gpu id parsing, ``semantic_nc``)
without touching CUDA device state.  Flags of reference subsystems that are out
of scope (datasets, VAE encoder, instance maps, ...) are accepted and ignored so
that a reference command line still parses.
"""
import argparse
import datetime
import os

ALIGN_CORNERS = False  # data/args.py:228 -- used by flow warp (models/utils.py:118,124)


def int_tuple(s):
    return tuple(int(i) for i in s.split(','))


def bool_flag(s):
    if s == '1':
        return True
    if s == '0':
        return False
    raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


def build_parser():
    p = argparse.ArgumentParser(description="AG2Video MI355X training step")
    a = p.add_argument
    a('--dataset', default='cater', choices=['smth_else', 'cater', 'synthetic', 'epic'])
    # optimisation
    a('--graph_only', default=0, type=int)
    a('--batch_size', default=3, type=int)
    a('--num_iterations', default=1000000, type=int)
    a('--learning_rate', default=1e-4, type=float)
    a('--beta1', default=0.5, type=float)
    a('--image_size', default='256,256', type=int_tuple)
    a('--num_val_samples', default=64, type=int)
    a('--loader_num_workers', default=4, type=int)
    # graph nets
    a('--mask_size', default=0, type=int)
    a('--embedding_dim', default=128, type=int)
    a('--gconv_dim', default=128, type=int)
    a('--g_mask_dim', default=128 + 64, type=int)
    a('--mask_noise_dim', default=64, type=int)
    a('--gconv_hidden_dim', default=512, type=int)
    a('--gconv_pooling', default='avg', type=str)
    a('--gconv_num_layers', default=3, type=int)
    a('--mlp_normalization', default='none', type=str)
    a('--num_upsampling_layers', choices=('normal', 'more', 'most'), default='normal')
    a('--ngf', type=int, default=64)
    a('--num_D', type=int, default=2)
    a('--n_layers_D', type=int, default=4)
    a('--use_actions_loss', default=1, type=int)
    a('--layout_arch', default='graph', type=str)
    a('--only_temporal', default=0, type=int)
    # flow generator
    a('--n_blocks_F', type=int, default=6)
    a('--nff', type=int, default=32)
    a('--n_downsample_F', type=int, default=3)
    a('--flow_deconv', action='store_true')
    a('--flow_multiplier', type=int, default=20)
    # temporal
    a('--frames_per_action', default=4, type=int)
    a('--frames_per_action_graph', default=4, type=int)
    a('--n_frames_G', type=int, default=2)
    a('--n_frames_D', type=int, default=1)
    # output
    a('--print_every', default=10, type=int)
    a('--timing', default=False, type=bool_flag)
    a('--checkpoint_every', default=10000, type=int)
    a('--output_dir', default=os.path.join(os.getcwd(), 'output',
                                           "{:%Y%m%dT%H%M}_%s".format(datetime.datetime.now())))
    a('--run_name', default="debug")
    a('--restore_checkpoint', default=0, type=int)
    a('--freeze_graph', action='store_true')
    a('--use_cuda', default=1, type=int)
    a('--gpu_ids', type=str, default='0')
    a('--norm_G', type=str, default='spectralspadesyncbatch3x3')
    a('--norm_D', type=str, default='spectralinstance')
    a('--norm_E', type=str, default='spectralinstance')
    a('--norm_F', type=str, default='spectralsyncbatch')
    a('--aspect_ratio', type=float, default=1.0)
    a('--init_type', type=str, default='xavier')
    a('--init_variance', type=float, default=0.02)
    # losses
    a('--ndf', type=int, default=64)
    a('--no_ganFeat_loss', action='store_true')
    a('--no_vgg_loss', action='store_true')
    a('--gan_mode', type=str, default='hinge')
    a('--lambda_feat', type=float, default=10.0)
    a('--lambda_vgg', type=float, default=10.0)
    a('--lambda_F_warp', type=float, default=10.0)
    a('--discriminator_img_loss_weight', default=1.0, type=float)
    a('--bbox_pred_loss_weight', default=10, type=float)
    a('--bp_prev', default=0, type=int)
    a('--gan_loss_type', default='gan')
    a('--debug', action='store_true')
    # build-specific (not in the reference): synthetic data shape for scripts/train
    a('--synthetic_objects', default='5,10', type=int_tuple, help='min,max real objects per clip')
    a('--seed', default=1234, type=int)
    a('--sync_bn', default=-1, type=int, help='batch-norm statistics over the global batch (all ranks) with the reference\'s multi-device formula (sync_batchnorm/batchnorm.py:78-93): 1 on, 0 off (per-rank statistics, F.batch_norm formula), -1 (default) = what the reference does at this world size: on when more than one rank runs, off on a single device')
    a('--data_root', default=None, type=str, help='CATER directory (scenes/*.json, videos/<id>/*.png) or, with --dataset smth_else, a Something-Else '
      'directory (offical_release_boxes/{objs_mapping.json, boxes_by_video_id/<id>/boxes.pkl}, frames/...); default: synthetic clips')
    a('--train_list', default=None, type=str, help='CATER: split list (default <data_root>/lists/train.txt); smth_else: label table (default <data_root>/train.csv)')
    a('--data_nframes', default=301, type=int, help='frames per video in --data_root')
    a('--vgg_weights', default=None, type=str, help='torchvision vgg19 state dict (.pth) for the perceptual loss')
    a('--temporal_D', default=0, type=int, help='1: add the temporal multiscale PatchGAN (reference discriminator.py:16-87: the frames of a '
      'window stacked into the channels of a 2-D PatchGAN; the reference constructs it nowhere) with hinge + feature-matching terms '
      'over non-overlapping windows of frames_per_action generated frames (BASELINE "3-D PatchGAN" row D4)')
    a('--lean_tape', default=0, type=int, help='1: smaller autograd tape for a few extra passes over memory (ops.set_lean_tape: compact gamma copies '
      'in the SPADE backward, the layout convolution recomputed for its batch norm): c3 138.6 -> ~105 GB peak, ~ +1.5 %% step time')
    a('--vgg_random_init', action='store_true', help='explicit opt-in: run the perceptual loss on a RANDOMLY initialised VGG19 (throughput / parity runs; not a training objective)')
    return p


parser = build_parser()


def init_args(args):
    """data/args.py:184-207: resolve gpu id list and semantic_nc; needs args.vocab."""
    try:
        args.output_dir = args.output_dir % args.run_name
    except Exception:
        pass
    if getattr(args, 'debug', False):
        args.num_val_samples = 20
    if isinstance(args.gpu_ids, str):
        ids = []
        for s in args.gpu_ids.split(','):
            i = int(s)
            if i >= 0:
                ids.append(i)
        args.gpu_ids = ids
    assert len(args.gpu_ids) == 0 or args.batch_size % len(args.gpu_ids) == 0, \
        "Batch size %d is wrong. It must be a multiple of # GPUs %d." % (args.batch_size, len(args.gpu_ids))
    args.semantic_nc = len(args.vocab['attributes']) * args.embedding_dim
    return args


def get_args(argv=None):
    args, _unknown = parser.parse_known_args(argv)
    return args


def make_opt(vocab, argv=()):
    """Convenience: parsed flags + vocab + init_args, as scripts/train.py:337-346 leaves them."""
    args, _ = parser.parse_known_args(list(argv))
    args.vocab = vocab
    return init_args(args)
