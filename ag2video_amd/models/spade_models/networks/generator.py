"""Layout -> video generator (reference: networks/generator.py:11-92)."""
import os

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


def _flow_precision():
    """Arithmetic of the flow network (see _frames_sequential).  AG2V_FLOW_FWD = exact | bf16 selects the two alternatives the
    margin test measures against it (tests/test_configs_gpu.py::test_c3prime_flow_forward_margin)."""
    mode = os.environ.get('AG2V_FLOW_FWD', 'h16')
    if mode == 'exact':
        return ops.bf16x3_min_rows(ops.EXACT, 32)
    if mode == 'bf16':
        return ops.bf16x3_min_rows(32, 32)
    return ops.bf16x3_min_rows(32, 32, fwd_h16=True)


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
        B, Fr = layout.size(0), layout.size(1)
        att = self.attribute_embedding(objs)                                     # [B, O, 384]
        vecs = torch.cat([att.unsqueeze(1).expand(B, Fr, att.size(1), att.size(2)), obj_vecs], dim=-1)
        # The reference renders every layout into [B, F+1, 512, H, W] (18 GB at 256x256x16f, batch 8) and convolves it densely
        # twice per frame step.  A layout is rank <= n_objects, so both first convolutions are evaluated in low-rank form
        # (layout_conv_lowrank) and the layouts are never materialised.  low_rank=False keeps the dense route.
        low_rank = getattr(opt, 'low_rank_layout', True)
        keep = remove_dummy_objects(objs, opt.vocab)
        nc = self.n_frames_G - 1
        # Teacher forcing (training: every frame pass reads GROUND-TRUTH previous frames, generator.py:71): the F - nc passes do
        # not depend on each other and run as one frame-major batch.  test_mode / bp_prev feed generated frames back: sequential.
        batched = self._takes_batched_path(Fr, test_mode)
        if batched:
            gen, flows, confs = self._frames_batched(imgs_gt, vecs, layout, keep, nc)
        else:
            gen, flows, confs = self._frames_sequential(imgs_gt, vecs, layout, keep, nc, objs, low_rank, test_mode)
        imgs = torch.stack(gen, dim=1)
        # the reference writes slot t-1 for t in [nc, Fr) into zero tensors (generator.py:57-58,78-79): nc-1 leading zero slots,
        # the flows, one trailing zero slot
        zf, zc = torch.zeros_like(flows[0]), torch.zeros_like(confs[0])
        flow_prev = torch.stack([zf] * (nc - 1) + flows + [zf], dim=1)
        conf_prev = torch.stack([zc] * (nc - 1) + confs + [zc], dim=1)
        return imgs, flow_prev, conf_prev

    def _takes_batched_path(self, Fr, test_mode):
        opt = self.opt
        return bool(getattr(opt, 'low_rank_layout', True) and not test_mode and not opt.bp_prev and Fr - (self.n_frames_G - 1) > 1
                    and getattr(opt, 'batch_frames', True))

    def _run_sn_bank(self, T):
        """The T power iterations of ALL spectral-norm layers, three launches per iteration (every layer is called exactly once per frame
        pass in the reference, so each consumes its T scales in the frame-batched pass)."""
        if getattr(self, '_sn_bank', None) is None:
            from .normalization import SpectralConv2d
            self._sn_bank = ops.SpectralBank([m for m in self.modules() if isinstance(m, SpectralConv2d)])
        for m, sc in zip(self._sn_bank.layers, self._sn_bank.scales(T)):
            m._bank_scales = sc

    def prefetch_spectral_scales(self, Fr, test_mode=False):
        """Round 6: the power iterations depend on the weights only (~8 ms of memory-bound kernels per c3 step).  AG2VideoModel.forward
        calls this BEFORE it waits for the appearance net's program, so they run beside the graph-net programs instead of behind them;
        the frame-batched pass of the SAME forward call then finds them done.  No-op whenever that pass would not run them itself (the
        iterations advance weight_u / weight_v: they must happen exactly once per step)."""
        self._sn_ready_T = None
        if self.training and torch.is_grad_enabled() and self._takes_batched_path(Fr, test_mode):
            T = Fr - (self.n_frames_G - 1)
            self._run_sn_bank(T)
            self._sn_ready_T = T

    def _frames_batched(self, imgs_gt, vecs, layout, keep, nc):
        """All T = F - nc teacher-forced frame passes as ONE pass over the frame-major batch [T*B, ...] (sample t*B + b = pass
        t + nc of clip b).  Per-pass quantities of the reference stay per frame group (ops.frame_groups): batch-norm statistics
        and running-stat updates, spectral norm's power iterations / sigma_t.  Same arithmetic per sample as the sequential loop;
        what changes is the launch count (1/T) and the GEMM shapes (T times the pixel columns: no split-K on the 8x8 ... 32x32
        layers, one weight pack / power-iteration chain / weight-gradient launch per layer and step instead of T)."""
        opt = self.opt
        H = opt.image_size[0]
        B, Fr = layout.size(0), layout.size(1)
        T = Fr - nc

        def fm(x, f):       # frames f .. f+T-1 of x [B, F, ...] -> frame-major [T*B, ...]
            return x[:, f:f + T].transpose(0, 1).reshape((T * B,) + tuple(x.shape[2:]))
        vecs_list = [fm(vecs, f) for f in range(nc + 1)]         # layouts of frames t-nc .. t
        boxes_list = [fm(layout, f) for f in range(nc + 1)]
        keep_t = keep.unsqueeze(0).expand(T, B, keep.size(1)).reshape(T * B, keep.size(1))
        prev = [fm(imgs_gt, f) for f in range(nc)]               # ground-truth frames t-nc .. t-1
        conv, bn = self.conv_dim_in[0][0], self.conv_dim_in[0][1]
        if self.training:
            # the T power iterations of ALL spectral-norm layers up front -- unless this call's prefetch has run them already
            ready, self._sn_ready_T = getattr(self, '_sn_ready_T', None), None
            if ready != T:
                self._run_sn_bank(T)
        with ops.frame_groups(T):
            spec = (vecs_list, boxes_list, keep_t, H)
            with _flow_precision():                               # forward on the fp16 split, backward on the bf16 one: see _frames_sequential
                _, flow = self.flows_network(layout=spec, prev=prev[0] if nc == 1 else torch.cat(prev, dim=1))
            warp = resample(prev[-1], flow)
            d = prev[-1] - warp
            conf = (torch.sum(d * d, dim=1, keepdim=True) < 0.02).float()
            with ops.bf16x3_min_rows(48), ops.timing_scope('conv_dim_in'):      # (netG opens its own scope inside)
                w, scales = conv.weight_and_scales()
                nseg = w.shape[1] - 3
                y = layout_conv_lowrank(w[:, :nseg], vecs_list, boxes_list, keep_t, H, H, conv.stride, conv.padding,
                                        sample_scale=scales.repeat_interleave((T * B) // scales.numel()),
                                        img=warp, w_img=w[:, nseg:], gscale=scales)
                # the normalised layout features are read by convolutions only (every SPADE's mlp_shared at its resolution, netG.fc):
                # stored as split words when all of them run on the bf16x3 kernels (their producers then skip the fp32 split)
                spl = self.netG.layout_split_ok(y.shape[0], y.shape[1], H, H)
                sizes = sorted(set(self.netG.pyramid_sizes(H, H)) | {(H, H)})
                if ops.batchnorm_act_pyramid_supported(y.shape, sizes):
                    # batch norm + the decoder's segmentation-map pyramid as one node: its backward forms the sum of the level
                    # gradients while the norm backward loads it
                    from .normalization import SegmapPyramid
                    lv = bn.forward_pyramid(y, sizes, slope=0.2, out_split=spl)
                    x = SegmapPyramid.from_levels(lv[(H, H)], lv)
                else:
                    x = bn(y, slope=0.2, out_split=spl)
                y = None        # (lean tape: the batch norm did not keep its input -- this name must not either)
                out = self.netG(x) + warp
        out = out.view(T, B, out.size(1), H, H)
        flow = flow.view(T, B, 2, H, H)
        conf = conf.view(T, B, 1, H, H)
        gen = [imgs_gt[:, i] for i in range(nc)] + list(out.unbind(0))
        return gen, list(flow.unbind(0)), list(conf.unbind(0))

    def _frames_sequential(self, imgs_gt, vecs, layout, keep, nc, objs, low_rank, test_mode):
        opt = self.opt
        H = opt.image_size[0]
        Fr = layout.size(1)
        segs = None if low_rank else render_clip_layouts(vecs, layout, objs, opt.vocab, H, H)
        gen = [imgs_gt[:, i] for i in range(nc)]
        confs, flows = [], []
        conv, bn = self.conv_dim_in[0][0], self.conv_dim_in[0][1]
        for t in range(nc, Fr):
            if test_mode or opt.bp_prev:
                prev = gen[-nc:]
            else:
                prev = [imgs_gt[:, i] for i in range(t - nc, t)]
            # Flow network: its output feeds the warp, which turns a flow error into an image error of |grad image| x error: the 1e-5
            # of the bf16 split products became 1.3e-3 of max|imgs_pred| against the oracle at the default widths (c1) and 3.4-4.1e-3
            # at 256 x 256 (limit 1e-3; tests/test_configs_gpu.py::test_c3prime_flow_forward_margin).  Rounds 1-2 ran its FORWARD on the
            # exact fp32 tiles (108 TF).  Round 3: the fp16 3-term split -- 11 + 11 significand bits instead of 8 + 8, ~5e-6 per
            # product like an fp32 fmaf chain (tools/ubench/f16x3_probe.hip) at the bf16x3 rate; its operands are batch-normalised
            # activations and O(0.03) filters (packed x 32), the range where fp16's narrow exponent costs nothing.  Gradients are not
            # amplified: the backward stays on the bf16 split from 32 rows on.
            with _flow_precision():
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
            # Image decoder: 64-channel layers on the bf16x3 tiles too (the flow network above stays exact in its forward).
            with ops.bf16x3_min_rows(48), ops.timing_scope('conv_dim_in'):      # (netG opens its own scope inside)
                if low_rank and conv.training:       # the frame-batched pass's formulation with one group (SpectralConv2d.forward)
                    w, scales = conv.weight_and_scales()
                    nseg = w.shape[1] - 3
                    y = layout_conv_lowrank(w[:, :nseg], spec[0], spec[1], keep, H, H, conv.stride, conv.padding,
                                            sample_scale=scales.repeat_interleave(warp.shape[0] // scales.numel()),
                                            img=warp, w_img=w[:, nseg:], gscale=scales)
                elif low_rank:
                    w = conv.effective_weight()
                    nseg = w.shape[1] - 3
                    y = layout_conv_lowrank(w[:, :nseg], spec[0], spec[1], keep, H, H, conv.stride, conv.padding,
                                            img=warp, w_img=w[:, nseg:])
                else:
                    y = conv(torch.cat(seg_t + [warp], dim=1))
                x = bn(y, slope=0.2)
                gen.append(self.netG(x) + warp)
        return gen, flows, confs
