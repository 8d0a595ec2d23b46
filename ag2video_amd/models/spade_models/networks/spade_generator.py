"""SPADE decoder (reference: networks/spade_generator.py:8-81)."""
import torch.nn as nn

from .... import ops
from ...layers import Conv2d
from .architecture import SPADEResnetBlock
from .normalization import SegmapPyramid


class SPADEGenerator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nf = opt.ngf
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        self.fc = Conv2d(opt.semantic_nc, 16 * nf, 3, padding=1)
        self.head_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt)
        self.up_2 = SPADEResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEResnetBlock(2 * nf, 1 * nf, opt)
        final_nc = nf
        if opt.num_upsampling_layers == 'most':
            self.up_4 = SPADEResnetBlock(1 * nf, nf // 2, opt)
            final_nc = nf // 2
        self.conv_img = Conv2d(final_nc, 3, 3, padding=1)

    @staticmethod
    def compute_latent_vector_size(opt):
        n = {'normal': 5, 'more': 6, 'most': 7}.get(opt.num_upsampling_layers)
        if n is None:
            raise ValueError('opt.num_upsampling_layers [%s] not recognized' % opt.num_upsampling_layers)
        sw = opt.image_size[0] // (2 ** n)
        sh = round(sw / opt.aspect_ratio)
        return sw, sh

    def pyramid_sizes(self, H, W):
        """Resolutions at which forward() evaluates SPADE layers: (sh, sw) doubled once per up-sampling."""
        n_up = {'normal': 5, 'more': 6, 'most': 7}[self.opt.num_upsampling_layers]
        return [(self.sh << i, self.sw << i) for i in range(n_up + 1)]

    def layout_split_ok(self, N, C, H, W):
        """Whether the layout features [N, C, H, W] may be handed to forward() as split words (ops.is_split): every reader -- the
        mlp_shared convolution of each SPADE at each resolution and the first convolution self.fc -- must take them in its forward
        pass and its weight gradient under the arithmetic settings in force (ops.conv_split_input_ok)."""
        from .normalization import SPADE
        shared = set()
        for m in self.modules():
            if isinstance(m, SPADE):
                conv = m.mlp_shared[0]
                shared.add((tuple(conv.weight.shape), int(conv.stride[0] if isinstance(conv.stride, tuple) else conv.stride),
                            int(conv.padding[0] if isinstance(conv.padding, tuple) else conv.padding)))
        for (h, w) in self.pyramid_sizes(H, W):
            for wshape, st, pad in shared:
                if not ops.conv_split_input_ok((N, C, h, w), wshape, st, pad):
                    return False
        fc = self.fc
        wshape = tuple((fc.weight_orig if hasattr(fc, 'weight_orig') else fc.weight).shape)
        return ops.conv_split_input_ok((N, C, self.sh, self.sw), wshape, 1, 1)

    def forward(self, layout):
        with ops.timing_scope('netG'):          # bench.py's per-stack roofline (ops.kernel_timing_summary(by_scope=True))
            return self._forward_impl(layout)

    def _forward_impl(self, layout):
        if not isinstance(layout, SegmapPyramid):      # (the frame-batched generator hands over the pyramid it built with the batch norm)
            layout = SegmapPyramid(layout, self.pyramid_sizes(layout.shape[2], layout.shape[3]))
        x = layout.at(self.sh, self.sw)
        x = self.fc(x)
        x = self.head_0(x, layout)
        x = ops.upsample2x(x)
        x = self.G_middle_0(x, layout)
        if self.opt.num_upsampling_layers in ('more', 'most'):
            x = ops.upsample2x(x)
        x = self.G_middle_1(x, layout)
        for blk in (self.up_0, self.up_1, self.up_2, self.up_3):
            # nn.Upsample(scale_factor=2) + block: the block reads x at its own resolution (and is x's only reader)
            x = blk(x, layout, up=2, x_single_reader=True)
        if self.opt.num_upsampling_layers == 'most':
            x = self.up_4(x, layout, up=2)
        x = self.conv_img(ops.leaky_relu(x, 0.2))
        return ops.tanh(x)
