"""SPADE residual block (reference: networks/architecture.py:21-68)."""
import os

import torch.nn as nn

from .... import ops

from .normalization import SPADE, SpectralConv2d


class SPADEResnetBlock(nn.Module):
    def __init__(self, fin, fout, opt):
        super().__init__()
        self.learned_shortcut = (fin != fout)
        fmiddle = min(fin, fout)
        assert 'spectral' in opt.norm_G, "norm_G without spectral norm is not on the hot path"
        self.conv_0 = SpectralConv2d(fin, fmiddle, 3, padding=1)
        self.conv_1 = SpectralConv2d(fmiddle, fout, 3, padding=1)
        if self.learned_shortcut:
            self.conv_s = SpectralConv2d(fin, fout, 1, padding=0, bias=False)
        cfg = opt.norm_G.replace('spectral', '')
        self.norm_0 = SPADE(cfg, fin, opt.semantic_nc)
        self.norm_1 = SPADE(cfg, fmiddle, opt.semantic_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE(cfg, fin, opt.semantic_nc)

    def forward(self, x, seg, up=1, x_single_reader=False):
        """up = 2: ``x`` is the block input BEFORE the nn.Upsample(scale_factor=2) that precedes this block in SPADEGenerator
        (spade_generator.py:62-76); the block then works at twice x's resolution.
        x_single_reader: the caller reads x nowhere else (its gradient may then come back as split words: ops.spade_pair)."""
        # norm_0, norm_1 (and norm_s) see the segmentation map at the same resolution (conv_0 keeps the size): their mlp_shared
        # convolutions run as one shared-input group, whose input gradient is accumulated in place (ops.conv2d_shared_input)
        norms = [self.norm_0, self.norm_1] + ([self.norm_s] if self.learned_shortcut else [])
        sh = [m.mlp_shared[0] for m in norms]
        H, W = x.shape[2] * up, x.shape[3] * up
        seg_r = seg.at(H, W) if hasattr(seg, 'at') else ops.nearest_resize(seg, H, W)
        # each activation is read by its SPADE's gamma | beta convolution only (forward, weight gradient, ReLU mask of the data
        # gradient): written as split words by the convolution epilogue where that convolution takes them
        N_ = x.shape[0]
        aspl = [self.training and ops.conv_split_input_ok((N_, c.out_channels, H, W), (2 * m.mlp_gamma.out_channels, c.out_channels) +
                                                          tuple(m.mlp_gamma.kernel_size), m.mlp_gamma.stride[0], m.mlp_gamma.padding[0])
                for m, c in zip(norms, sh)]
        # ... and, when all of those convolutions can, their data gradients (which apply this ReLU's mask) hand d(actv) back split
        gspl = self.training and all(
            ops.conv_split_dy_ok((N_, c.out_channels, H, W), (2 * m.mlp_gamma.out_channels, c.out_channels) + tuple(m.mlp_gamma.kernel_size),
                                 m.mlp_gamma.stride[0], m.mlp_gamma.padding[0]) for m, c in zip(norms, sh))
        actv = ops.conv2d_shared_input(seg_r, [c.weight for c in sh], [c.bias for c in sh], sh[0].stride[0], sh[0].padding[0],
                                       relu=True, grad_premasked=True, out_split=aspl, grad_split=gspl)
        bn0, bns = self.norm_0.param_free_norm, (self.norm_s.param_free_norm if self.learned_shortcut else None)
        # the modulated tensors are read by one convolution each and nothing else: stored as split words where that convolution
        # takes them (forward + weight gradient on the bf16x3 kernels, ops.conv_split_input_ok)

        def takes_split(conv, cin):
            return self.training and ops.conv_split_input_ok((x.shape[0], cin, H, W), conv.weight_orig.shape, conv.stride, conv.padding)
        s0 = takes_split(self.conv_0, x.shape[1])
        ss = self.learned_shortcut and takes_split(self.conv_s, x.shape[1])
        s1 = takes_split(self.conv_1, self.conv_0.out_channels)
        from .... import distributed as D
        if (self.learned_shortcut and self.training and ops.spade_pair_supported(H, W, up)
                and os.environ.get('AG2V_NO_SPADE_PAIR') is None):
            # norm_0 and norm_s modulate the SAME tensor: one statistics pass, one read of x for both outputs, one summed input
            # gradient -- and with up = 2 the upsampled tensor never exists (ops.spade_pair)
            y0, ys = ops.spade_pair(x, self.norm_0.gamma_beta(actv[0]), self.norm_s.gamma_beta(actv[2]), bn0.running_mean,
                                    bn0.running_var, bns.running_mean, bns.running_var, up, bn0.momentum, bn0.eps, 0.2, 1.0,
                                    ops.current_frame_groups(), out_split=(s0, ss), xs_single_reader=x_single_reader)
            x_s = self.conv_s(ys)
            dx = self.conv_0(y0)
        else:
            if up == 2:
                x = ops.upsample2x(x)
            x_s = self.conv_s(self.norm_s(x, seg, actv=actv[2], out_split=ss)) if self.learned_shortcut else x
            dx = self.conv_0(self.norm_0(x, seg, slope=0.2, actv=actv[0], out_split=s0))
        # (dx = conv_0's output is read by norm_1 and nothing else: its gradient may come back as split words)
        # (with a learned shortcut x_s = conv_s(...) is read here and nowhere else: a split-word d(out) passes through to it)
        return self.conv_1(self.norm_1(dx, seg, slope=0.2, actv=actv[1], out_split=s1, x_single_reader=True),
                           residual=x_s, residual_single_reader=self.learned_shortcut)                  # out = x_s + dx fused in the conv epilogue

# torchvision.models.vgg19 "features" configuration ('E': 16 conv3x3 + ReLU layers, 'M' = MaxPool2d(2, 2)); the reference pins
# torchvision 0.5.0 (README.md:23).  Module indices inside ``features`` are what the slices below are named after.
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def vgg19_feature_layers():
    """[(index in torchvision's vgg19().features, kind, in_channels, out_channels)] for kind in conv / relu / pool."""
    layers, c_in, idx = [], 3, 0
    for v in VGG19_CFG:
        if v == 'M':
            layers.append((idx, 'pool', c_in, c_in))
            idx += 1
        else:
            layers.append((idx, 'conv', c_in, v))
            layers.append((idx + 1, 'relu', v, v))
            c_in, idx = v, idx + 2
    return layers


class _ReLUMarker(nn.Module):
    """Placeholder for features[i] = ReLU: the activation runs in the preceding convolution's epilogue."""

    def forward(self, x):
        return x


class _MaxPool2(nn.Module):
    def forward(self, x):
        return ops.maxpool2(x)


class VGG19(nn.Module):
    """The five feature slices of VGG19 used by the perceptual loss (reference: architecture.py:93-123): relu1_1, relu2_1, relu3_1,
    relu4_1, relu5_1 = features[0:2], [2:7], [7:12], [12:21], [21:30].  Children are named by their index in torchvision's
    ``features`` (hence state_dict keys ``slice3.10.weight`` ...), so a torchvision VGG19 ``features`` state dict maps onto this module
    key by key (load_torchvision_features).  The reference loads ImageNet weights from the network; there is none here, so the
    weights are whatever the caller fills in (closed-form random init for throughput / parity runs).  Frozen, as in the reference."""

    SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))

    def __init__(self, requires_grad=False):
        super().__init__()
        from ...layers import Conv2d
        layers = {i: (kind, ci, co) for i, kind, ci, co in vgg19_feature_layers()}
        for s, (lo, hi) in enumerate(self.SLICES):
            seq = nn.Sequential()
            for i in range(lo, hi):
                kind, ci, co = layers[i]
                mod = Conv2d(ci, co, kernel_size=3, padding=1) if kind == 'conv' else (_ReLUMarker() if kind == 'relu' else _MaxPool2())
                seq.add_module(str(i), mod)
            setattr(self, 'slice%d' % (s + 1), seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_features(self, features_state_dict):
        """features_state_dict: torchvision vgg19().features.state_dict() ('0.weight', '0.bias', '2.weight', ...)."""
        own = self.state_dict()
        for k, v in features_state_dict.items():
            idx = int(k.split('.')[0])
            s = next(i for i, (lo, hi) in enumerate(self.SLICES) if lo <= idx < hi) + 1
            own['slice%d.%s' % (s, k)].copy_(v)

    def forward(self, X):
        out, h = [], X
        for s in range(1, 6):
            for mod in getattr(self, 'slice%d' % s):
                if isinstance(mod, nn.Conv2d):
                    h = ops.conv2d(h, mod.weight, mod.bias, 1, 1, relu=True)       # conv + the ReLU that follows it
                elif isinstance(mod, _MaxPool2):
                    h = mod(h)
            out.append(h)
        return out
