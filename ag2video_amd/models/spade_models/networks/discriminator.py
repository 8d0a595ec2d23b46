"""Action-conditioned multiscale PatchGAN discriminator
(reference: networks/discriminator.py:212-399, the only discriminator classes training instantiates)."""
import numpy as np
import torch
import torch.nn as nn

from .... import ops
from ...attribute_embed import AttributeEmbeddings
from ...graph_models.graph import GraphTripleConv
from ...graph_models.model import action_edge_vectors
from ...layers import Conv2d, Embedding, FusedMLP, Linear
from ...layout import layout_conv_lowrank, render_clip_layouts
from ...utils import remove_dummy_objects
from .normalization import get_nonspade_norm_layer


class NLayerActionDiscriminator(nn.Module):
    """PatchGAN: conv4x4 s2 + lrelu; (SN-conv4x4 + InstanceNorm + lrelu) x (n_layers_D-1); conv4x4 -> 1 channel.
    Children are model0..model{n}: same nesting (hence state_dict keys) as discriminator.py:357-384."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        kw = 4
        padw = int(np.ceil((kw - 1.0) / 2))
        nf = opt.ndf
        input_nc = self.compute_D_input_nc()
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        sequence = [[Conv2d(input_nc, nf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, False)]]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            sequence += [[norm_layer(nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=stride, padding=padw)),
                          nn.LeakyReLU(0.2, False)]]
        sequence += [[Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)]]
        for n in range(len(sequence)):
            self.add_module('model' + str(n), nn.Sequential(*sequence[n]))

    def compute_D_input_nc(self):
        emb = self.opt.gconv_dim if self.opt.use_actions_loss else self.opt.semantic_nc
        return emb * 2 + 3

    def forward(self, x=None, first_conv=None):
        """x: the 259-channel input; or first_conv: the pre-activation output of model0's convolution computed by the caller
        (MultiscaleActionDiscriminator evaluates its layout channels in low-rank form)."""
        results = []
        for idx, sub in enumerate(self.children()):
            first = sub[0]
            if isinstance(first, nn.Sequential):            # SN conv -> instance norm (+ fused LeakyReLU)
                x = first[1](first[0](x), slope=0.2)
            else:
                x = first_conv if (idx == 0 and first_conv is not None) else first(x)
                if len(sub) > 1:                            # model0: conv + LeakyReLU
                    x = ops.leaky_relu(x, 0.2)
            results.append(x)
        if not self.opt.no_ganFeat_loss:
            return results
        return results[-1]


class NLayerDiscriminatorTM(NLayerActionDiscriminator):
    """Temporal PatchGAN scale (reference: discriminator.py:16-59): the same model0..model{n} stack over the clip's frames stacked
    into channels -- input_nc = frames_per_action * 3 (:46-47)."""

    def compute_D_input_nc(self):
        return self.opt.frames_per_action * 3


class MultiscaleDiscriminatorT(nn.Module):
    """Temporal multiscale discriminator (reference: discriminator.py:62-87): img [B, T, 3, H, W] -> view [B, T*3, H, W] -> num_D
    PatchGANs with 3x3 / stride-2 average pooling between scales.  BASELINE.json's "3-D PatchGAN": the reference has no 3-D
    convolution on any path and constructs this class nowhere; it is its only discriminator that sees time."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, NLayerDiscriminatorTM(opt))

    def downsample(self, x):
        return ops.avgpool3s2(x)

    def forward(self, img):
        bs, nt, ch, h, w = img.size()
        x = img.reshape(bs, ch * nt, h, w)
        result = []
        with ops.bf16x3_min_rows(48), ops.timing_scope('D_tmp'):
            for name, D in self.named_children():
                if name.startswith('discriminator'):
                    out = D(x)
                    result.append(out if not self.opt.no_ganFeat_loss else [out])
                    x = self.downsample(x)
        return result


class MultiscaleActionDiscriminator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, NLayerActionDiscriminator(opt))
        emb = opt.embedding_dim
        self.attribute_embedding = AttributeEmbeddings(opt.vocab['attributes'], emb)
        self.pred_embeddings = Embedding(len(opt.vocab['pred_idx_to_name']), emb)
        self.acts_embeddings = Embedding(len(opt.vocab['action_idx_to_name']), emb)
        num_attributes = len(opt.vocab['attributes'].keys())
        obj_input_dim = num_attributes * emb
        first = dict(obj_input_dim=obj_input_dim, object_output_dim=opt.gconv_dim, predicate_input_dim=emb,
                     predicate_output_dim=opt.gconv_dim, hidden_dim=opt.gconv_hidden_dim, num_attributes=num_attributes,
                     mlp_normalization=opt.mlp_normalization, pooling=opt.gconv_pooling, loc_dim=4)
        general = dict(first, obj_input_dim=first["object_output_dim"], predicate_input_dim=opt.gconv_dim)
        self.gconvs = nn.ModuleList([GraphTripleConv(**first), GraphTripleConv(**general)])
        self.obj_vecs_net = FusedMLP(Linear(emb + 4, obj_input_dim, bias=False), nn.ReLU(inplace=True),
                                     Linear(obj_input_dim, obj_input_dim, bias=False), nn.ReLU())
        self.pre_obj_vecs_net = FusedMLP(Linear(obj_input_dim, emb, bias=False), nn.ReLU(inplace=True),
                                         Linear(emb, emb, bias=False), nn.ReLU())
        self.fc_objs_vecs = Linear(opt.gconv_dim + opt.semantic_nc, opt.gconv_dim * 2)

    def downsample(self, x):
        return ops.avgpool3s2(x)

    def get_obj_vecs(self, objs, layout_boxes, actions_data):
        """Recurrent 2-layer graph net over action edges only (discriminator.py:273-313)."""
        triplets, temporal_triplets, rel, locs = actions_data
        x_end, y_end = locs[:, :, :, 0], locs[:, :, :, 1]
        pad = self.opt.vocab["action_name_to_idx"]["__padding__"]
        ov = self.pre_obj_vecs_net(self.attribute_embedding(objs))
        out = []
        # edge lists, masks and action-edge vectors of ALL time steps at once, frame-major (the recurrence only carries ov)
        s, a, o = temporal_triplets[..., 0], temporal_triplets[..., 1], temporal_triplets[..., 2]           # [B, T, A]
        edges_all = torch.stack([s, o], dim=-1).long().transpose(0, 1).contiguous()
        ind_all = (a != pad).transpose(0, 1).contiguous()
        pv_fm = action_edge_vectors(self.acts_embeddings, a, x_end, y_end, rel).transpose(0, 1).contiguous()         # [T, B, A, D]
        boxes_steps = layout_boxes.transpose(0, 1).contiguous()
        from .... import graph_program
        if ov.is_cuda and graph_program.supported(self.opt, T=pv_fm.shape[2]):
            lin = [m for m in self.obj_vecs_net.children() if isinstance(m, Linear)]
            out, _ = graph_program.run_recurrence('ov', ov, pv_fm, edges_all, ind_all, boxes_steps, lin, list(self.gconvs), None)
            return out.permute(1, 0, 2, 3)
        pv_steps = pv_fm.unbind(0)
        for t in range(layout_boxes.size(1)):
            ov = self.obj_vecs_net(torch.cat([ov, boxes_steps[t]], dim=-1))
            edges, ind, pv = edges_all[t], ind_all[t], pv_steps[t]
            for gconv in self.gconvs:
                ov, pv = gconv(ov, pv, edges, ind)
            out.append(ov)
        return torch.stack(out, dim=1)

    def forward(self, img, objs, layout_boxes, actions_data, layout_terms=None):
        with ops.timing_scope('D'):          # bench.py's per-stack roofline (ops.kernel_timing_summary(by_scope=True))
            return self._forward_impl(img, objs, layout_boxes, actions_data, layout_terms=layout_terms)

    def _forward_impl(self, img, objs, layout_boxes, actions_data, layout_terms=None):
        """img [B,T,3,H,W], layout_boxes [B,T,O,4] -> num_D lists of n_layers_D+1 feature maps over [B*T, ...].
        ``layout_terms``: the result of self.layout_terms(objs, layout_boxes, actions_data), when the caller evaluates the
        discriminator on several images with the same layout (LossModel: fake and real)."""
        with ops.bf16x3_min_rows(48):      # the 64-channel first layers too (no warp downstream: ops.bf16x3_min_rows)
            return self._forward(img, objs, layout_boxes, actions_data, layout_terms)

    def _low_rank(self):
        return getattr(self.opt, 'low_rank_layout', True) and self.opt.num_D <= 2

    def layout_terms(self, objs, layout_boxes, actions_data):
        """The part of every scale's first convolution that comes from the 256 layout channels (graph net on the actions ->
        object vectors -> low-rank layout convolution).  It depends on the layout inputs and this module's parameters only, not
        on the image: the reference recomputes it inside each of its four discriminator passes per step
        (loss_model.py:73,77,118,120); LossModel computes it once per loss and shares it between the fake and the real pass
        (identical values; in the discriminator loss autograd sums the two passes' gradients into the shared node).
        Returns None when the dense route is configured."""
        if not self._low_rank():
            return None
        with ops.bf16x3_min_rows(48), ops.timing_scope('D'):
            H = self.opt.image_size[0]
            ov = self.get_obj_vecs(objs, layout_boxes, actions_data)
            att = self.attribute_embedding(objs)
            B, T = ov.size(0), ov.size(1)
            v = self.fc_objs_vecs(torch.cat([att.unsqueeze(1).expand(B, T, att.size(1), att.size(2)), ov], dim=-1))
            keep = remove_dummy_objects(objs, self.opt.vocab)                                   # [B, O]
            keep = keep.unsqueeze(1).expand(B, T, keep.size(1)).reshape(B * T, -1)
            vf = v.reshape(B * T, v.size(2), v.size(3))
            bf = layout_boxes.reshape(B * T, layout_boxes.size(2), 4)
            terms, scale = [], 0
            for name, D in self.named_children():
                if name.startswith('discriminator'):
                    conv = D.model0[0]
                    terms.append(layout_conv_lowrank(conv.weight[:, 3:], [vf], [bf], keep, H, H, conv.stride[0], conv.padding[0],
                                                     residual=None, pool=scale))
                    scale += 1
            return terms

    def _forward(self, img, objs, layout_boxes, actions_data, layout_terms=None):
        H = self.opt.image_size[0]
        result = []
        if self._low_rank():
            # The 256 layout channels of the 259-channel input are a sum over objects of (vector) x (box weight map): both
            # scales' first convolutions take them in low-rank form (scale 1 with the pooled box profiles), so neither the
            # [B*T, 259, H, W] input nor its pooled copy is ever built; only the 3 image channels are convolved densely, with
            # the layout term added in the convolution's epilogue.
            if layout_terms is None:
                layout_terms = self.layout_terms(objs, layout_boxes, actions_data)
            B, T = img.size(0), img.size(1)
            x_img = img.reshape(B * T, img.size(2), H, H)
            scale = 0
            for name, D in self.named_children():
                if name.startswith('discriminator'):
                    conv = D.model0[0]
                    nimg = x_img.size(1)
                    y = ops.conv2d(x_img, conv.weight[:, :nimg], conv.bias, conv.stride[0], conv.padding[0],
                                   residual=layout_terms[scale])
                    out = D(first_conv=y)
                    result.append(out if not self.opt.no_ganFeat_loss else [out])
                    x_img = self.downsample(x_img)
                    scale += 1
            return result
        ov = self.get_obj_vecs(objs, layout_boxes, actions_data)
        att = self.attribute_embedding(objs)
        B, T = ov.size(0), ov.size(1)
        v = self.fc_objs_vecs(torch.cat([att.unsqueeze(1).expand(B, T, att.size(1), att.size(2)), ov], dim=-1))
        segs = render_clip_layouts(v, layout_boxes, objs, self.opt.vocab, H, H, frame_major=True)   # T x [B, 256, H, W]
        seg = torch.stack(segs, dim=1)                                                              # [B, T, 256, H, W]
        x = torch.cat([img, seg], dim=2)
        x = x.view(B * T, x.size(2), H, H)
        for name, D in self.named_children():
            if name.startswith('discriminator'):
                out = D(x)
                result.append(out if not self.opt.no_ganFeat_loss else [out])
                x = self.downsample(x)
        return result
