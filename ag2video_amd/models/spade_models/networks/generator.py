"""Layout -> video generator (reference: networks/generator.py:11-92)."""
import torch
import torch.nn as nn

from .... import ops
from ...attribute_embed import AttributeEmbeddings
from ...layout import layout_conv_lowrank, render_clip_layouts
from ...utils import remove_dummy_objects
from ...utils import resample
from .flows_generator import FlowsGenerator
from .normalization import get_nonspade_norm_layer
from .spade_generator import SPADEGenerator


class Layout2VidGenerator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.attribute_embedding = AttributeEmbeddings(opt.vocab['attributes'], 384 // len(opt.vocab['attributes']))
        self.netG = SPADEGenerator(opt)
        self.bs = opt.batch_size
        self.n_frames_G = opt.n_frames_G
        self.flows_network = FlowsGenerator(opt)
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_F)
        self.conv_dim_in = nn.Sequential(
            norm_layer(nn.Conv2d((opt.gconv_dim * 4) * self.n_frames_G + 3, opt.semantic_nc, kernel_size=3, padding=1)),
            nn.LeakyReLU(0.2, True))

    def forward(self, imgs_gt, objs, obj_vecs, layout, imgs_prev=None, test_mode=False):
        """imgs_gt [B,F,3,H,W], objs [B,O,A], obj_vecs [B,F,O,128], layout (boxes) [B,F,O,4] ->
        (imgs [B,F,3,H,W] with frame 0 = GT, flows [B,F,2,H,W], conf [B,F,1,H,W])"""
        opt = self.opt
        H = opt.image_size[0]
        B, Fr = layout.size(0), layout.size(1)
        att = self.attribute_embedding(objs)                                     # [B, O, 384]
        vecs = torch.cat([att.unsqueeze(1).expand(B, Fr, att.size(1), att.size(2)), obj_vecs], dim=-1)
        # The reference renders every layout into [B, F+1, 512, H, W] (18 GB at 256x256x16f, batch 8) and convolves it densely
        # twice per frame step.  A layout is rank <= n_objects, so both first convolutions are evaluated in low-rank form
        # (layout_conv_lowrank) and the layouts are never materialised.  low_rank=False keeps the dense route.
        low_rank = getattr(opt, 'low_rank_layout', True)
        keep = remove_dummy_objects(objs, opt.vocab)
        segs = None if low_rank else render_clip_layouts(vecs, layout, objs, opt.vocab, H, H)
        nc = self.n_frames_G - 1
        gen = [imgs_gt[:, i] for i in range(nc)]
        confs, flows = [], []
        conv, bn = self.conv_dim_in[0][0], self.conv_dim_in[0][1]
        for t in range(nc, Fr):
            if test_mode or opt.bp_prev:
                prev = gen[-nc:]
            else:
                prev = [imgs_gt[:, i] for i in range(t - nc, t)]
            if low_rank:
                frames = list(range(t - nc, t + 1))
                spec = ([vecs[:, f] for f in frames], [layout[:, f] for f in frames], keep, H)
                _, flow = self.flows_network(layout=spec, prev=prev[0] if nc == 1 else torch.cat(prev, dim=1))
            else:
                seg_t = segs[t - nc:t + 1]
                _, flow = self.flows_network(torch.cat(seg_t + prev, dim=1))
            warp = resample(prev[-1], flow)
            d = prev[-1] - warp
            confs.append((torch.sum(d * d, dim=1, keepdim=True) < 0.02).float())
            flows.append(flow)
            # Image decoder: 64-channel layers on the bf16x3 tiles too.  The flow network above keeps the library default
            # (128 rows): its 64-channel tail feeds the warp, which amplifies the 1e-5 split error (ops.bf16x3_min_rows).
            with ops.bf16x3_min_rows(48):
                if low_rank:
                    w = conv.effective_weight()
                    nseg = w.shape[1] - 3
                    y = ops.conv2d(warp, w[:, nseg:], None, conv.stride, conv.padding)
                    y = layout_conv_lowrank(w[:, :nseg], spec[0], spec[1], keep, H, H, conv.stride, conv.padding, residual=y)
                else:
                    y = conv(torch.cat(seg_t + [warp], dim=1))
                x = bn(y, slope=0.2)
                gen.append(self.netG(x) + warp)
        imgs = torch.stack(gen, dim=1)
        pad = Fr - len(flows)      # slots t-1 are filled; the trailing slot(s) stay zero (generator.py:57-58,78-79)
        flow_prev = torch.stack(flows + [torch.zeros_like(flows[0])] * pad, dim=1)
        conf_prev = torch.stack(confs + [torch.zeros_like(confs[0])] * pad, dim=1)
        return imgs, flow_prev, conf_prev
