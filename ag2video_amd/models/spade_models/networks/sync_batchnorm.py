"""Data-parallel glue with the reference's names (reference: networks/sync_batchnorm/{replicate,batchnorm}.py).

The reference is single-process torch.nn.DataParallel (threads + NCCL broadcast of every parameter every forward,
replicate.py:50-67).  This build runs one process per GPU: the wrapper below only reproduces the ``.module`` attribute
(hence the ``.module.`` infix of every state_dict key); gradient averaging is done once per step by
ag2video_amd.distributed over RCCL.  Batch-norm statistics: by default per process -- the single-device formula F.batch_norm
((var+eps)^-1/2, batchnorm.py:63-68), which the golden vectors pin; with ``--sync_bn 1`` (distributed.set_sync_bn) and live
collectives they are taken over the GLOBAL batch with the reference's multi-device formula (one all-reduce of [sum, sum of squares]
per call, inv_std = clamp(biased var, eps)^-1/2, unbiased variance into running_var: batchnorm.py:80-93,105-145;
distributed.combine_bn_stats), and the backward all-reduces its per-channel sums between the reduce and the apply pass.
"""
import torch
import torch.nn as nn

from .... import ops


class DataParallelWithCallback(nn.Module):
    def __init__(self, module, device_ids=None):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids else []

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class SynchronizedBatchNorm2d(nn.Module):
    """Batch norm over (N,H,W) with optional affine, fused with the LeakyReLU that follows it at every call site.
    Parameter / buffer names are those of torch.nn.BatchNorm2d.  num_batches_tracked is registered but never advanced,
    exactly like the reference's forward (it calls F.batch_norm directly, batchnorm.py:63-68)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x, slope=1.0, out_split=False):
        # inside ops.frame_groups(T): statistics per frame group, T running-stat updates in order (see ops.frame_groups)
        # out_split: the result as split words for bf16x3 convolutions (ops.batchnorm_act)
        return ops.batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.momentum, self.eps, slope, groups=ops.current_frame_groups(), out_split=out_split)

    def forward_pyramid(self, x, sizes, slope=1.0, out_split=False):
        """forward() followed by the nearest-neighbour pyramid of the result (every SPADE's resized segmentation map) as one autograd
        node (ops.batchnorm_act_pyramid); returns {size: tensor}."""
        outs = ops.batchnorm_act_pyramid(x, self.weight, self.bias, self.running_mean, self.running_var, sizes, self.training,
                                         self.momentum, self.eps, slope, groups=ops.current_frame_groups(), out_split=out_split)
        return {tuple(int(v) for v in k): t for k, t in zip(sizes, outs)}

    def modulate(self, x, gamma, beta, slope=1.0):
        """SPADE use: normalise without affine and apply (1+gamma), beta in the same pass."""
        return ops.spade_norm(x, gamma, beta, self.running_mean, self.running_var, self.training, self.momentum, self.eps,
                              slope, groups=ops.current_frame_groups())

    def modulate_fused(self, x, gb, slope=1.0, out_split=False, x_single_reader=False):
        """modulate() with gamma | beta as the channel halves of one [N, 2C, H, W] tensor.  x_single_reader: see ops.spade_norm_fused."""
        return ops.spade_norm_fused(x, gb, self.running_mean, self.running_var, self.training, self.momentum, self.eps, slope,
                                    groups=ops.current_frame_groups(), out_split=out_split, x_single_reader=x_single_reader)
