"""Normalisation layers (reference: networks/normalization.py:16-110)."""
import re

import torch
import torch.nn as nn

from .... import ops
from ...layers import Conv2d
from .sync_batchnorm import SynchronizedBatchNorm2d


class SpectralConv2d(nn.Module):
    """Conv2d under torch.nn.utils.spectral_norm: parameters ``weight_orig`` (+ ``bias``), buffers ``weight_u`` /
    ``weight_v``; every training-mode forward advances the power iteration by one step and convolves with
    weight_orig / sigma (normalization.py:26-27, architecture.py:35-39, flows_generator.py:79-85)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, eps=1e-12):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.eps = kernel_size, stride, padding, eps
        ref = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)  # default init
        self.weight_orig = nn.Parameter(ref.weight.detach().clone())
        if bias:
            self.bias = nn.Parameter(ref.bias.detach().clone())
        else:
            self.register_parameter('bias', None)
        w = self.weight_orig
        u = torch.nn.functional.normalize(w.new_empty(out_channels).normal_(0, 1), dim=0, eps=eps)
        v = torch.nn.functional.normalize(w.new_empty(w.numel() // out_channels).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)

    def drop_bias(self):
        """get_nonspade_norm_layer removes the bias of a conv that feeds a norm layer (normalization.py:35-37)."""
        self.bias = None

    def effective_weight(self):
        return ops.spectral_weight(self.weight_orig, self.weight_u, self.weight_v, self.training, self.eps)

    def weight_and_scales(self):
        """Frame-batched form: (weight_orig, [1 / sigma_t for the T frame groups of ops.frame_groups]) -- T power iterations
        in a row, as the T forward calls of the reference's frame loop make them; conv2d(..., gscale=scales) applies them."""
        pending = getattr(self, '_bank_scales', None)
        if pending is not None:              # produced for every layer of the network at once (ops.SpectralBank, generator.py)
            self._bank_scales = None
            return self.weight_orig, pending
        return self.weight_orig, ops.spectral_scales(self.weight_orig, self.weight_u, self.weight_v, ops.current_frame_groups(),
                                                     self.training, self.eps)

    def forward(self, x, residual=None, residual_single_reader=False):
        # Training: ONE formulation for the frame-batched pass (T groups) and a single call (T = 1) -- the raw weight_orig in the
        # GEMM and 1 / sigma_t applied to the product in the epilogue.  (conv(x, W / sigma) and conv(x, W) / sigma differ in the last
        # bit; with both paths on the same form the frame-batched and the sequential generator agree bit for bit in deterministic
        # mode, tests/test_fullsize_gpu.py.)  Eval: W / sigma without a power iteration.
        if self.training or ops.current_frame_groups() > 1:
            w, scales = self.weight_and_scales()
            return ops.conv2d(x, w, self.bias, self.stride, self.padding, residual, gscale=scales, residual_single_reader=residual_single_reader)
        return ops.conv2d(x, self.effective_weight(), self.bias, self.stride, self.padding, residual)


class InstanceNormLReLU(nn.Module):
    """nn.InstanceNorm2d(affine=False) fused with the LeakyReLU(0.2) that follows it in the discriminator
    (normalization.py:43-44, discriminator.py:374-377).  No parameters, no buffers."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def forward(self, x, slope=0.2):
        return ops.instnorm_lrelu(x, self.eps, slope)


def get_nonspade_norm_layer(opt, norm_type='instance'):
    """Returns add_norm_layer(conv_module) -> nn.Sequential(spectral conv, norm) (normalization.py:16-50).
    ``layer`` is described by its constructor arguments (a torch.nn.Conv2d instance is accepted and re-created)."""

    def add_norm_layer(layer):
        assert norm_type.startswith('spectral'), "only spectral* norm types are on the hot path (data/args.py:103-109)"
        sub = norm_type[len('spectral'):]
        conv = SpectralConv2d(layer.in_channels, layer.out_channels, layer.kernel_size[0], stride=layer.stride[0],
                              padding=layer.padding[0], bias=layer.bias is not None)
        if sub == 'none' or len(sub) == 0:
            return conv
        conv.drop_bias()
        if sub in ('batch', 'sync_batch', 'syncbatch'):
            norm = SynchronizedBatchNorm2d(layer.out_channels, affine=True)
        elif sub == 'instance':
            norm = InstanceNormLReLU(layer.out_channels)
        else:
            raise ValueError('normalization layer %s is not recognized' % sub)
        return nn.Sequential(conv, norm)

    return add_norm_layer


class SegmapPyramid(object):
    """F.interpolate(segmap, size, 'nearest') for every SPADE of one decoder pass (normalization.py:102).  Each SPADE of the
    reference resizes the same map again; here the map is resized once per size, and -- when ``sizes`` lists the decoder's
    resolutions up front -- all sizes belong to one autograd node whose backward writes the map's gradient in a single pass
    (ops.nearest_pyramid) instead of one full-size scatter and one full-size accumulation per size."""

    def __init__(self, segmap, sizes=None):
        self.segmap = segmap
        self._cache = {}
        if sizes:
            sizes = sorted(set((int(h), int(w)) for (h, w) in sizes))
            for key, t in zip(sizes, ops.nearest_pyramid(segmap, sizes)):
                self._cache[key] = t

    @classmethod
    def from_levels(cls, segmap, levels):
        """A pyramid whose levels were built elsewhere ({(h, w): tensor}; SynchronizedBatchNorm2d.forward_pyramid)."""
        p = cls(segmap)
        p._cache.update(levels)
        return p

    def at(self, h, w):
        key = (int(h), int(w))
        if key not in self._cache:
            self._cache[key] = ops.nearest_resize(self.segmap, key[0], key[1])
        return self._cache[key]


class SPADE(nn.Module):
    """Spatially-adaptive normalisation (normalization.py:66-110): x_hat * (1 + gamma(seg)) + beta(seg)."""

    def __init__(self, config_text, norm_nc, label_nc):
        super().__init__()
        assert config_text.startswith('spade')
        parsed = re.search(r'spade(\D+)(\d)x\d', config_text)
        kind, ks = str(parsed.group(1)), int(parsed.group(2))
        if kind not in ('syncbatch', 'batch'):
            raise ValueError('%s is not a param-free norm on the hot path (norm_G default spadesyncbatch3x3)' % kind)
        self.param_free_norm = SynchronizedBatchNorm2d(norm_nc, affine=False)
        nhidden = 128
        pw = ks // 2
        self.mlp_shared = nn.Sequential(Conv2d(label_nc, nhidden, kernel_size=ks, padding=pw), nn.ReLU())
        self.mlp_gamma = Conv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)
        self.mlp_beta = Conv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)

    def resized_segmap(self, x, segmap):
        if isinstance(segmap, SegmapPyramid):
            return segmap.at(x.shape[2], x.shape[3])
        return ops.nearest_resize(segmap, x.shape[2], x.shape[3])

    def forward(self, x, segmap, slope=1.0, actv=None, out_split=False, x_single_reader=False):
        """slope: LeakyReLU fused behind the modulation (the caller's actvn, architecture.py:53-54); 1.0 = none.
        ``segmap`` may be a SegmapPyramid: every SPADE of a decoder pass resizes the same map to one of six sizes, so
        the nearest-neighbour resize (normalization.py:102) is done once per size instead of once per SPADE.
        ``actv``: relu(mlp_shared(resized segmap)) computed by the caller together with the other SPADEs that read the same map
        (SPADEResnetBlock: ops.conv2d_shared_input).  ``x_single_reader``: nothing but this SPADE reads x (ops.spade_norm_fused)."""
        if actv is None:
            seg = self.resized_segmap(x, segmap)
            # mlp_shared = Conv2d + ReLU: the ReLU rides in the convolution's epilogue and its backward in the epilogue of the
            # gamma|beta convolution's data gradient (actv has no other consumer)
            sh = self.mlp_shared[0]
            actv = ops.conv2d(seg, sh.weight, sh.bias, sh.stride[0], sh.padding[0], relu=True, grad_premasked=True)
        return self.param_free_norm.modulate_fused(x, self.gamma_beta(actv), slope, out_split=out_split, x_single_reader=x_single_reader)

    def gamma_beta(self, actv):
        """gamma | beta as ONE [N, 2C, H, W] tensor: mlp_gamma and mlp_beta read the same activation, so they run as one convolution
        with 2C output rows (a GEMM with twice the rows for the same operand traffic, one data gradient instead of two plus their
        sum) whose output the modulation kernels read in place.  The parameters stay separate (state_dict names of the reference)."""
        w, b = self._fused_gamma_beta()
        return ops.conv2d(actv, w, b, self.mlp_gamma.stride[0], self.mlp_gamma.padding[0], relu_input=True)

    def _fused_gamma_beta(self):
        """cat(mlp_gamma, mlp_beta) weights / biases, rebuilt when a parameter changed (optimizer step, load_state_dict) or the
        autograd mode did; within one training step every frame's forward shares the node (one cat backward per step)."""
        ps = (self.mlp_gamma.weight, self.mlp_beta.weight, self.mlp_gamma.bias, self.mlp_beta.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps) + (ops.param_epoch(), torch.is_grad_enabled())
        if getattr(self, '_gb_key', None) != key:
            w, b = torch.cat([ps[0], ps[1]], dim=0), torch.cat([ps[2], ps[3]], dim=0)
            if w.requires_grad:
                w.register_hook(self._drop_fused)      # its graph is consumed by the backward: rebuild for the next forward
            self._gb_cache, self._gb_key = (w, b), key
        return self._gb_cache

    def _drop_fused(self, grad):
        self._gb_key = None
        return None
