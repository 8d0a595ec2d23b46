"""Layout -> video generator (reference: networks/generator.py:11-92)."""
import torch
import torch.nn as nn

from .... import ops
from ...attribute_embed import AttributeEmbeddings
from ...layout import render_clip_layouts
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
        # one launch renders every (sample, frame) layout; frame f is segs[f] [B, 512, H, W].  (The reference also appends
        # a copy of the last layout, generator.py:53; no frame step ever reads it.)
        segs = render_clip_layouts(vecs, layout, objs, opt.vocab, H, H)
        nc = self.n_frames_G - 1
        gen = [imgs_gt[:, i] for i in range(nc)]
        confs, flows = [], []
        conv, bn = self.conv_dim_in[0][0], self.conv_dim_in[0][1]
        for t in range(nc, Fr):
            seg_t = segs[t - nc:t + 1]
            if test_mode or opt.bp_prev:
                prev = gen[-nc:]
            else:
                prev = [imgs_gt[:, i] for i in range(t - nc, t)]
            _, flow = self.flows_network(torch.cat(seg_t + prev, dim=1))
            warp = resample(prev[-1], flow)
            d = prev[-1] - warp
            confs.append((torch.sum(d * d, dim=1, keepdim=True) < 0.02).float())
            flows.append(flow)
            x = bn(conv(torch.cat(seg_t + [warp], dim=1)), slope=0.2)
            gen.append(self.netG(x) + warp)
        imgs = torch.stack(gen, dim=1)
        pad = Fr - len(flows)      # slots t-1 are filled; the trailing slot(s) stay zero (generator.py:57-58,78-79)
        flow_prev = torch.stack(flows + [torch.zeros_like(flows[0])] * pad, dim=1)
        conf_prev = torch.stack(confs + [torch.zeros_like(confs[0])] * pad, dim=1)
        return imgs, flow_prev, conf_prev
