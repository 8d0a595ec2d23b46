"""SPADE residual block (reference: networks/architecture.py:21-68)."""
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

    def forward(self, x, seg):
        # norm_0, norm_1 (and norm_s) see the segmentation map at the same resolution (conv_0 keeps the size): their mlp_shared
        # convolutions run as one shared-input group, whose input gradient is accumulated in place (ops.conv2d_shared_input)
        norms = [self.norm_0, self.norm_1] + ([self.norm_s] if self.learned_shortcut else [])
        sh = [m.mlp_shared[0] for m in norms]
        seg_r = self.norm_0.resized_segmap(x, seg)
        actv = ops.conv2d_shared_input(seg_r, [c.weight for c in sh], [c.bias for c in sh], sh[0].stride[0], sh[0].padding[0],
                                       relu=True, grad_premasked=True)
        x_s = self.conv_s(self.norm_s(x, seg, actv=actv[2])) if self.learned_shortcut else x
        dx = self.conv_0(self.norm_0(x, seg, slope=0.2, actv=actv[0]))
        return self.conv_1(self.norm_1(dx, seg, slope=0.2, actv=actv[1]), residual=x_s)   # out = x_s + dx fused in the conv epilogue
