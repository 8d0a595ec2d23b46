"""SPADE residual block (reference: networks/architecture.py:21-68)."""
import torch.nn as nn

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
        x_s = self.conv_s(self.norm_s(x, seg)) if self.learned_shortcut else x
        dx = self.conv_0(self.norm_0(x, seg, slope=0.2))
        return self.conv_1(self.norm_1(dx, seg, slope=0.2), residual=x_s)   # out = x_s + dx fused in the conv epilogue
