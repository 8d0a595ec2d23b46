"""End-to-end gradient check of the training step, layer by layer ("a 1 % systematic gradient error in one layer
would pass every step test").

The step tests compare gradients with the oracle's at 2e-2 ... 5e-2: the fp32 generator is ill-conditioned (one ReLU decision moves a
gradient by 1e-3 ... 1e-2), so a tighter bound against ANOTHER implementation's rounding is not available.  This test needs no second
implementation: for one parameter tensor at a time it moves the tensor along its own gradient direction d = g / |g| and compares the
central difference of the GPU step's loss, (L(theta + eps d) - L(theta - eps d)) / (2 eps), with |g| -- the product's forward pass
against the product's backward pass, through every kernel between that tensor and the functional.  A decoder or discriminator
layer whose gradient is off by 1 % (a dropped term, a wrong scale, a mask applied twice) fails the 5e-3 bound (measured worst:
2.6e-3 in exact fp32, 4.7e-3 in the bf16x3 arithmetic under an 8e-3 bound); tensors in front of the whole generator (flow network,
graph nets, embeddings) are held to 2e-2 (measured 1.3e-2 ... 1.5e-2), limited by the roughness of an fp32 forward pass as a function
of parameters that every activation depends on.  The functional is a fixed random linear form of the step's outputs accumulated in
fp64 -- the step's own L1 / hinge losses put a kink wherever two features cross, which a finite difference smears over (with them the
same check reads 2e-2 ... 4e-1); the loss kernels have their own tests (tests/test_ops_gpu.py::test_losses).  Two widths: the toy
networks (16-channel base) and the default ones (c1: 64-channel base, 512-wide graph nets -- the 256 x 128 / 128 x 256 tiles,
the graph-net programs at their real shapes).  The run is deterministic: it reproduces bit for bit.

What makes the two sides the same function:
* deterministic reductions and the exact-fp32 convolution kernels (no split-K order noise in L);
* spectral norm treats u, v as constants in the backward (as torch.nn.utils.spectral_norm does), but every forward pass re-runs the
  power iteration, so u, v DO follow the weights in a finite difference.  At the fixed point of the iteration that dependence
  vanishes to first order (u, v are singular vectors: d sigma = u^T dW v), so the test first runs the step without updates until the
  vectors have converged, and restores those buffers before every evaluation;
* batch-norm running statistics do not enter a training-mode forward pass.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

FLAGS = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '2', '--ngf', '16', '--ndf', '16', '--nff', '16',
         '--gconv_hidden_dim', '64', '--gpu_ids', '0']
# parameter tensors of the generator side: (substring of the name, which network) -- one per kind of kernel path
G_TENSORS = [
    'layout_to_video.module.netG.fc.weight',                              # first decoder convolution, on the smallest pyramid level (split-word input)
    'layout_to_video.module.netG.head_0.conv_0.weight_orig',              # SPADE block convolution, 1/sigma_t in the epilogue
    'layout_to_video.module.netG.head_0.norm_0.mlp_shared.0.weight',      # shared-input convolution group (merged dgrad / wgrad)
    'layout_to_video.module.netG.up_1.norm_0.mlp_gamma.weight',           # gamma half of the fused gamma | beta convolution (SPADE pair)
    'layout_to_video.module.netG.up_1.norm_0.mlp_beta.bias',              # a bias of it: channel sums from the modulation's reduce pass
    'layout_to_video.module.netG.up_3.conv_1.weight_orig',                # 64-row layer with the residual in the epilogue
    'layout_to_video.module.netG.up_3.conv_s.weight_orig',                # 1 x 1 learned shortcut
    'layout_to_video.module.netG.conv_img.weight',                        # 3-row output convolution (few-row kernels)
    'layout_to_video.module.conv_dim_in.0.0.weight_orig',                 # low-rank layout convolution + image channels
    'layout_to_video.module.conv_dim_in.0.1.weight',                      # affine batch norm behind it (pyramid node)
    'layout_to_video.module.flows_network.down_flow.0.0.weight_orig',     # flow network: layout convolution, stride
    'layout_to_video.module.flows_network.res_flow.0.conv_1.weight_orig', # residual block (fp16-split forward / bf16 backward pair in the product; exact here)
    'layout_to_video.module.flows_network.res_flow.0.bn_0.weight',        # affine batch norm + ReLU inside it
    'layout_to_video.module.flows_network.conv_flow.0.weight',            # 2-row flow head, through the warp
    'layout_to_video.module.attribute_embedding.att_emb_0.weight',        # embedding table -> layout taps
    'acts_to_objs.module.gconvs.1.net1.0.weight',                         # graph-net program (appearance net recurrence)
    'acts_to_objs.module.acts_embeddings.weight',
    # the box net (trained by the graph-only step; here through the probe of boxes_pred): its recurrence feeds each step's boxes back in
    'acts_to_boxes.module.gconvs.2.net2.0.weight',
    'acts_to_boxes.module.box_net.2.weight',
]
D_TENSORS = [
    'discriminator_0.model0.0.weight',                                    # first PatchGAN layer (no norm)
    'discriminator_0.model1.0.0.weight_orig',                             # spectral conv + instance norm
    'discriminator_1.model2.0.0.weight_orig',                             # second scale (avg-pooled input)
    'layout_terms',                                                       # (resolved below: a layout / action term tensor of the discriminator)
]


def _resolve(named, want):
    hits = [(n, p) for n, p in named if want in n]
    return hits[0] if hits else (None, None)


FLAGS_DEFAULT_WIDTH = ['--no_vgg_loss', '--image_size', '64,64', '--batch_size', '1', '--gpu_ids', '0']      # c1: ngf = ndf = 64, 512-wide graph nets


@pytest.mark.parametrize("width", ['toy', 'default'])
@pytest.mark.parametrize("prec", ['fp32', 'bf16x3'])
def test_loss_gradients_match_finite_differences_layer_by_layer(prec, width, device, monkeypatch):
    from ag2video_amd import ops
    from ag2video_amd.data.args import make_opt
    from ag2video_amd.data.synthetic import make_clip_batch
    from ag2video_amd.data.vocab import cater_vocab
    from ag2video_amd.engine import TrainStep
    if prec == 'fp32':
        monkeypatch.setenv('AG2V_FLOW_FWD', 'exact')       # (bf16x3: the product's arithmetic -- fp16-split flow forward, bf16-split backward)
    old_prec = ops.get_conv_precision()
    ops.set_conv_precision(prec)
    try:
        with ops.deterministic():
            opt = make_opt(cater_vocab(), FLAGS if width == 'toy' else FLAGS_DEFAULT_WIDTH)
            ts = TrainStep(opt, device, weights_seed=5)
            clip = ts.to_device(make_clip_batch(opt.vocab, opt.batch_size, 4, (64, 64), seed=11, max_objs=6))
            for _ in range(int(os.environ.get('AG2V_GRADCHECK_WARM', '60'))):      # power iterations to their fixed points (no updates)
                ts.gd_step(clip, apply_updates=False)
            bufs = [(b, b.detach().clone()) for m in (ts.model, ts.discriminator) for b in m.buffers()]

            def restore():
                for b, v in bufs:
                    b.copy_(v)

            lm = ts.gans_model.module
            nc = lm.n_frames_G - 1
            gen = torch.Generator(device='cpu').manual_seed(77)
            probes = {}

            def probe(key, t):
                """<R, t> in fp64 with a fixed random R ~ N(0, 1 / numel): a smooth functional of an output tensor (the L1 / hinge losses
                of the step put a kink wherever two features cross, which a finite difference smears over; the loss kernels have their own
                tests), accumulated in double so that the sum itself adds no rounding noise."""
                if key not in probes:
                    probes[key] = (torch.randn(t.shape, generator=gen) / float(t.numel()) ** 0.5).to(t.device).double()
                return (probes[key] * t.double()).sum()

            def g_loss(grad):
                restore()
                ctx = torch.enable_grad() if grad else torch.no_grad()
                with ctx:
                    out = ts.model(clip[0], clip[1], clip[3], clip[4], boxes_gt=clip[2], test_mode=False, use_gt=True)
                    L = probe('img', out[0][:, nc:]) * 4.0 + probe('flow', out[2]) + probe('boxes', out[1]) * 4.0
                return L, out

            def d_loss(grad, out):
                restore()
                ctx = torch.enable_grad() if grad else torch.no_grad()
                with ctx:
                    r_adata = [item[:, nc:] for item in out[4]]
                    feats = lm.netD_img(out[0][:, nc:], clip[1], clip[2][:, nc:], r_adata)
                    L = 0.0
                    for i, scale in enumerate(feats):
                        for j, f in enumerate(scale):
                            L = L + probe('d%d_%d' % (i, j), f)
                return L

            # ---- gradients (the training step's own backward passes)
            ts.optimizer_generator.zero_grad()
            ts.optimizer_graph.zero_grad()
            L0, out0 = g_loss(True)
            with ops.accumulate_into_param_grads():
                L0.backward()
            g_named = list(ts.model.named_parameters())
            picks = []
            for want in G_TENSORS:
                n, p = _resolve(g_named, want)
                assert p is not None and p.grad is not None, want
                picks.append(('G', n, p, p.grad.detach().clone()))
            out_det = tuple(t.detach() if torch.is_tensor(t) else t for t in out0)
            ts.discriminator.optimizer_d_img.zero_grad()
            LD0 = d_loss(True, out_det)
            with ops.accumulate_into_param_grads():
                LD0.backward()
            d_named = [(n, p) for n, p in ts.discriminator.img_discriminator.named_parameters() if p.grad is not None]
            for want in D_TENSORS:
                if want == 'layout_terms':
                    cand = [(n, p) for n, p in d_named if n.startswith('gconvs.') or 'embedding' in n or 'obj_vecs' in n]
                    n, p = max(cand, key=lambda np_: float(np_[1].grad.norm())) if cand else (None, None)
                else:
                    n, p = _resolve(d_named, want)
                assert p is not None, want
                picks.append(('D', n, p, p.grad.detach().clone()))

            # ---- central differences along each tensor's own gradient direction, at two step sizes.
            # * The networks are piecewise smooth (ReLU / LeakyReLU): the fraction of units that change branch grows with the step,
            #   which makes the difference quotient err linearly in eps -- removed by extrapolating the two quotients to eps = 0.
            # * Some tensors (flow network, graph nets: amplified by the warp of white-noise frames) have gradients so large that the
            #   step for the targeted change of the functional moves a weight by a few ulps.  The quotient therefore uses the step the
            #   fp32 parameters ACTUALLY took: predicted = <g, theta_plus - theta_minus>, exact differences of the stored values.
            # * Step sizes: the fp32 forward pass is a rough function of its parameters -- a perturbation changes the rounding of every
            #   activation behind it -- with an amplitude of ~3e-4 of these functionals when the perturbed tensor sits in front of the
            #   whole generator (flow network, graph nets, embeddings) and ~1e-5 for a decoder or discriminator layer.  The change of the
            #   functional per side is therefore 0.016 ... 0.024 for the latter (bound 5e-3) and 0.045 ... 0.09 for the former (bound
            #   2e-2); the extrapolated quotients of the step sizes are averaged.
            scale = float(os.environ.get('AG2V_GRADCHECK_SCALE', '1'))
            report, worst = [], {False: 0.0, True: 0.0}
            bad = []
            for side, n, p, g in picks:
                upstream = side == 'G' and any(k in n for k in ('flows_network', 'acts_to_objs', 'acts_to_boxes', 'attribute_embedding'))
                targets = (0.045, 0.064, 0.09) if upstream else (0.016, 0.024)
                # (bf16x3: the split products add their own 1e-5-class roughness -- measured worst 4.7e-3 / 1.5e-2, exact fp32 2.6e-3 / 1.3e-2)
                bound = (2e-2 if upstream else 5e-3) if prec == 'fp32' else (2.5e-2 if upstream else 8e-3)
                gn = float(g.double().norm())
                assert gn > 0, n
                d = (g / gn).to(p.dtype)
                p0 = p.detach().clone()
                q0s = []
                for target in targets:
                    qs = []
                    for eps in (scale * target / gn, 0.5 * scale * target / gn):
                        vals, thetas = [], []
                        for sgn in (+1.0, -1.0):
                            with torch.no_grad():
                                p.copy_(p0)
                                p.add_(d, alpha=sgn * eps)
                            thetas.append(p.detach().clone())
                            ops.bump_param_epoch()              # derived-tensor caches (fused gamma | beta weights) follow the parameters
                            vals.append(float(g_loss(False)[0]) if side == 'G' else float(d_loss(False, out_det)))
                        pred = float((g.double() * (thetas[0].double() - thetas[1].double())).sum())
                        qs.append((vals[0] - vals[1]) / pred)
                    q0s.append(2.0 * qs[1] - qs[0])
                with torch.no_grad():
                    p.copy_(p0)
                ops.bump_param_epoch()
                q0 = sum(q0s) / len(q0s)
                rel = abs(q0 - 1.0)
                line = '%s %-72s |g| %.3e  measured / predicted change -> %s  mean %.5f  rel %.2e (bound %.0e)' % (
                    side, n[-72:], gn, ' '.join('%.4f' % q for q in q0s), q0, rel, bound)
                report.append(line)
                worst[upstream] = max(worst[upstream], rel)
                if rel > bound:
                    bad.append(line)
            print('\n'.join(report))
            print('worst: decoder / discriminator tensors %.2e, tensors in front of the generator %.2e' % (worst[False], worst[True]))
            assert not bad, '\n'.join(bad)
    finally:
        ops.set_conv_precision(old_prec)
