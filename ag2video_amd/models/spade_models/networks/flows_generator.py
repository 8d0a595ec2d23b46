"""Optical-flow generator (reference: networks/flows_generator.py:14-109)."""
import contextlib
import os

import torch.nn as nn

from .... import ops
from ...layers import Conv2d
from ...layout import layout_conv_lowrank
from .normalization import SpectralConv2d, get_nonspade_norm_layer
from .sync_batchnorm import SynchronizedBatchNorm2d


class FlowResnetBlock(nn.Module):
    """The non-SPADE ``SPADEResnetBlock`` of flows_generator.py:71-109 with norm='spectralsyncbatch':
    BN -> lrelu -> SN-conv3x3 (twice) + identity shortcut (fin == fout, stride 1 at every call site)."""

    def __init__(self, fin, fout):
        super().__init__()
        assert fin == fout, "flow residual blocks are square (flows_generator.py:40-42)"
        fhidden = min(fin, fout)
        self.conv_0 = SpectralConv2d(fin, fhidden, 3, stride=1, padding=1)
        self.conv_1 = SpectralConv2d(fhidden, fout, 3, padding=1)
        self.bn_0 = SynchronizedBatchNorm2d(fin)
        self.bn_1 = SynchronizedBatchNorm2d(fhidden)

    def forward(self, x):
        dx = self.conv_0(self.bn_0(x, slope=0.2))
        return self.conv_1(self.bn_1(dx, slope=0.2), residual=x)


class FlowsGenerator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nc = opt.n_frames_G - 1
        input_nc = (opt.gconv_dim * 4) * opt.n_frames_G + nc * 3
        self.flow_multiplier = opt.flow_multiplier
        nf, nd = opt.nff, opt.n_downsample_F
        ch = [min(1024, nf * (2 ** i)) for i in range(nd + 1)]
        if opt.flow_deconv:
            raise NotImplementedError("--flow_deconv (ConvTranspose2d) is not the default path (data/args.py:74)")
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_F)
        act = lambda: nn.LeakyReLU(0.2, True)   # index placeholders: the activation is fused into the norm kernel
        down = [norm_layer(nn.Conv2d(input_nc, nf, kernel_size=3, padding=1)), act()]
        for i in range(nd):
            down += [norm_layer(nn.Conv2d(ch[i], ch[i + 1], kernel_size=3, padding=1, stride=2)), act()]
        ch_r = min(1024, nf * (2 ** nd))
        res = [FlowResnetBlock(ch_r, ch_r) for _ in range(opt.n_blocks_F)]
        up = []
        for i in reversed(range(nd)):
            up += [nn.Upsample(scale_factor=2), norm_layer(nn.Conv2d(ch[i + 1], ch[i], kernel_size=3, padding=1)), act()]
        self.down_flow = nn.Sequential(*down)
        self.res_flow = nn.Sequential(*res)
        self.up_flow = nn.Sequential(*up)
        self.conv_flow = nn.Sequential(Conv2d(nf, 2, kernel_size=3, padding=1))
        # conv_w: parameters exist in the reference state_dict; its output is computed and discarded there
        # (generator.py:75), so it receives no gradient and is not evaluated here.
        self.conv_w = nn.Sequential(Conv2d(nf, 1, kernel_size=3, padding=1), nn.Sigmoid())

    @staticmethod
    def _conv_bn_act(seq, x):
        return seq[1](seq[0](x), slope=0.2)

    def forward(self, label=None, layout=None, prev=None):
        with ops.timing_scope('flow'):          # bench.py's per-stack roofline (ops.kernel_timing_summary(by_scope=True))
            return self._forward_impl(label=label, layout=layout, prev=prev)

    def _forward_impl(self, label=None, layout=None, prev=None):
        """Either ``label`` = cat[layouts, previous frames] (dense first conv, as flows_generator.py:62-68), or
        ``layout`` = (vecs_list, boxes_list, keep, H) + ``prev`` [B, 3*nc, H, W]: the first conv then treats its layout
        channels in low-rank form (models/layout.py:layout_conv_lowrank) and only the image channels densely."""
        first = True
        x = label
        for m in self.down_flow:
            if isinstance(m, nn.Sequential):
                if first and layout is not None:
                    vecs_list, boxes_list, keep, H = layout
                    conv, bn = m[0], m[1]
                    T = ops.current_frame_groups()
                    if T > 1 or conv.training:      # raw weight, 1/sigma_t per frame group (SpectralConv2d.forward: one form in training)
                        w, scales = conv.weight_and_scales()
                        per_sample = scales.repeat_interleave(prev.shape[0] // scales.numel())
                    else:
                        w, scales, per_sample = conv.effective_weight(), None, None
                    nseg = w.shape[1] - prev.shape[1]
                    y = layout_conv_lowrank(w[:, :nseg], vecs_list, boxes_list, keep, H, H, conv.stride, conv.padding,
                                            sample_scale=per_sample, img=prev, w_img=w[:, nseg:], gscale=scales)
                    x = bn(y, slope=0.2)
                else:
                    x = self._conv_bn_act(m, x)
                first = False
        part = os.environ.get('AG2V_FLOW_SPLIT_PARTS', '')          # diagnostic (tests/debug/debug_c1_precision.py): forward of these parts on bf16x3
        ctx = lambda name: ops.bf16x3_min_rows(48, 32) if name in part else contextlib.nullcontext()
        with ctx('res'):
            for blk in self.res_flow:
                x = blk(x)
        with ctx('up'):
            for m in self.up_flow:
                if isinstance(m, nn.Upsample):
                    x = ops.upsample2x(x)
                elif isinstance(m, nn.Sequential):
                    x = self._conv_bn_act(m, x)
        flow = self.conv_flow[0](x) * self.flow_multiplier
        return None, flow
