"""ORACLE (test infrastructure, not product code): CPU restatement of every operator on the AG2Video hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  Each function restates,
in plain PyTorch-CPU fp32, the arithmetic of the reference call site it cites (file:line relative to the reference
checkout) with the same signature as the matching ag2video_amd.ops operator, so the parity tests read
``ops.f(*args) ~= oracle.kernels.f(*args)`` forward and (through autograd) backward.

Pinning: oracle/model.py composes these into the full training step, which tests/test_oracle_golden.py checks against
tests/golden/toy64_step.npz -- outputs of the unmodified reference modules run in the build container
(tests/golden/make_golden.py).  The tolerance of every comparison is stated in the test that makes it.
"""
import torch
import torch.nn.functional as F


def conv2d(x, w, bias=None, stride=1, pad=0, residual=None, relu=False, grad_premasked=False, relu_input=False, gscale=None,
           residual_single_reader=False):
    """nn.Conv2d.forward (aten.convolution), e.g. generator.py:30, architecture.py:29-32, discriminator.py:368-379.
    relu: the nn.ReLU that follows SPADE's mlp_shared convolution (normalization.py:92-95).  grad_premasked / relu_input only tell
    the product where to apply that ReLU's backward; autograd does it here.
    gscale [G]: the frame-batched form of a spectral-norm convolution -- sample group g (N / G consecutive samples = one frame pass
    of generator.py:62-87) is convolved with w * gscale[g], i.e. with that pass's weight_orig / sigma_g, one F.conv2d per pass."""
    if gscale is not None:
        G = gscale.numel()
        xs = x.reshape((G, x.shape[0] // G) + tuple(x.shape[1:]))
        y = torch.cat([F.conv2d(xs[g], w * gscale[g], None, stride=stride, padding=pad) for g in range(G)], dim=0)
        y = y if bias is None else y + bias.view(1, -1, 1, 1)
        y = y if residual is None else y + residual
        return F.relu(y) if relu else y
    y = F.conv2d(x, w, bias, stride=stride, padding=pad)
    y = y if residual is None else y + residual
    return F.relu(y) if relu else y


def conv2d_shared_input(x, weights, biases, stride=1, pad=0, relu=False, grad_premasked=False, out_split=None, grad_split=False):
    """Several convolutions of one input (SPADE.mlp_shared of norm_0 / norm_1 / norm_s on the same segmap, normalization.py:92-104)."""
    return tuple(conv2d(x, w, b, stride, pad, relu=relu) for w, b in zip(weights, biases))


def spectral_weight(w_orig, u, v, training=True, eps=1e-12):
    """torch.nn.utils.spectral_norm.SpectralNorm.compute_weight (n_power_iterations=1) as applied at
    architecture.py:35-39, flows_generator.py:79-85, normalization.py:26-27.  u, v are updated in place in training."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    uc, vc = u.clone(), v.clone()
    sigma = torch.dot(uc, torch.mv(wm, vc))
    return w_orig / sigma


def spectral_scales(w_orig, u, v, T, training=True, eps=1e-12):
    """[1 / sigma_t] of T consecutive train-mode forward calls of one spectral-norm layer (the T frame passes of
    generator.py:62-87): T times the power iteration + sigma of spectral_weight above, u / v advancing in place.  Eval: one value."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    out = []
    for _ in range(T if training else 1):
        if training:
            with torch.no_grad():
                v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
                u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        uc, vc = u.clone(), v.clone()
        out.append(1.0 / torch.dot(uc, torch.mv(wm, vc)))
    return torch.stack(out)


class SpectralBank(object):
    """Checker for ops.SpectralBank: the T power iterations of every layer, layer by layer (spectral_scales above)."""

    def __init__(self, layers):
        self.layers = list(layers)

    def scales(self, T, eps=1e-12):
        return tuple(spectral_scales(l.weight_orig, l.weight_u, l.weight_v, T, True, eps) for l in self.layers)


def _lrelu(y, slope):
    return y if slope == 1.0 else F.leaky_relu(y, slope)


# ---- the reference's MULTI-DEVICE batch norm (sync_batchnorm/batchnorm.py:80-93,105-126,128-145): every replica sums x and x^2 over
# its shard, the master adds them up (ReduceAddCoalesced), computes mean and inv_std = clamp(biased var, eps)^-1/2, feeds the
# running buffers with the UNBIASED variance and broadcasts (mean, inv_std); y = x * (inv_std w) + (b - mean inv_std w).  Gradients
# flow through the reduce / broadcast pair, which for one process per device is an all-reduce in both directions.
# SYNC_BN = True selects it (tests of --sync_bn 1); in a single process the all-reduce is the identity, i.e. the same formula over
# the whole local batch -- the single-process checker of a two-rank run on the two half batches.
SYNC_BN = False


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        import torch.distributed as dist
        t = t.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


def _sync_batch_norm(x, running_mean, running_var, weight, bias, momentum, eps):
    """_SynchronizedBatchNorm.forward, data-parallel branch (batchnorm.py:72-93) + _compute_mean_std (:128-145)."""
    import torch.distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    N, C = x.shape[0], x.shape[1]
    xr = x.reshape(N, C, -1)
    size = float(N * xr.shape[2] * world)
    packed = _AllReduceSum.apply(torch.stack([xr.sum(dim=(0, 2)), (xr * xr).sum(dim=(0, 2))]))
    ssum, sqsum = packed[0], packed[1]
    mean = ssum / size
    sumvar = sqsum - ssum * mean
    unbias_var, bias_var = sumvar / (size - 1), sumvar / size
    with torch.no_grad():
        running_mean.mul_(1 - momentum).add_(momentum * mean.detach())
        running_var.mul_(1 - momentum).add_(momentum * unbias_var.detach())
    inv_std = bias_var.clamp(min=eps) ** -0.5
    if weight is not None:
        scale = (inv_std * weight).view(1, C, 1)
        shift = (bias - mean * inv_std * weight).view(1, C, 1)
    else:
        scale, shift = inv_std.view(1, C, 1), (-mean * inv_std).view(1, C, 1)
    return (xr * scale + shift).reshape(x.shape)


def _batch_norm_groups(x, running_mean, running_var, weight, bias, training, momentum, eps, groups):
    """One F.batch_norm call per group of N / groups consecutive samples, in order (each call updates the running statistics):
    the frame passes of the reference's generator loop, held as one frame-major batch by the product."""
    if SYNC_BN and training:
        n = x.shape[0] // groups
        return torch.cat([_sync_batch_norm(x[g * n:(g + 1) * n], running_mean, running_var, weight, bias, momentum, eps)
                          for g in range(groups)], dim=0)
    if groups == 1 or not training:
        return F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)
    n = x.shape[0] // groups
    return torch.cat([F.batch_norm(x[g * n:(g + 1) * n], running_mean, running_var, weight, bias, training, momentum, eps)
                      for g in range(groups)], dim=0)


def conv_split_input_ok(x_shape, w_shape, stride, pad):
    """The product may store an activation as split bf16 words for its bf16x3 kernels (ops.is_split): a storage format of the GPU
    path with no counterpart in the reference's arithmetic -- the oracle always answers "keep fp32"."""
    return False


def batchnorm_act(x, weight, bias, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1,
                  out_split=False):
    """_SynchronizedBatchNorm single-device path = F.batch_norm (sync_batchnorm/batchnorm.py:63-68) + LeakyReLU
    (generator.py:26-33, flows_generator.py:30-54)."""
    assert not out_split
    return _lrelu(_batch_norm_groups(x, running_mean, running_var, weight, bias, training, momentum, eps, groups), slope)


def batchnorm_act_pyramid(x, weight, bias, running_mean, running_var, sizes, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1,
                          out_split=False):
    """The product's fused node = the reference's two steps: the normalised tensor (generator.py:26-33), then every SPADE's
    F.interpolate(segmap, size, 'nearest') of it (normalization.py:102)."""
    assert not out_split
    y = batchnorm_act(x, weight, bias, running_mean, running_var, training, momentum, eps, slope, groups)
    return nearest_pyramid(y, sizes)


def batchnorm_act_pyramid_supported(x_shape, sizes):
    return True


def spade_norm(x, gamma, beta, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1):
    """SPADE.forward normalization.py:99,108 (+ actvn architecture.py:53-54,68)."""
    n = _batch_norm_groups(x, running_mean, running_var, None, None, training, momentum, eps, groups)
    return _lrelu(n * (1 + gamma) + beta, slope)


def spade_norm_fused(x, gb, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, slope=1.0, groups=1, out_split=False,
                     x_single_reader=False):
    """spade_norm with gamma | beta stacked along channels (the product runs mlp_gamma and mlp_beta, normalization.py:105-106,
    as one convolution)."""
    C = x.shape[1]
    return spade_norm(x, gb[:, :C], gb[:, C:], running_mean, running_var, training, momentum, eps, slope, groups)


def spade_pair(xs, gb0, gb1, rm0, rv0, rm1, rv1, up=1, momentum=0.1, eps=1e-5, slope0=1.0, slope1=1.0, groups=1, out_split=(False, False),
               xs_single_reader=False):
    """Checker for ops.spade_pair: nn.Upsample(scale_factor=2) (spade_generator.py:62-76) followed by the block's two SPADE layers on
    the same input (norm_0 / norm_s, architecture.py:52-53,60-61), each with its own running buffers."""
    x = upsample2x(xs) if up == 2 else xs
    return (spade_norm_fused(x, gb0, rm0, rv0, True, momentum, eps, slope0, groups),
            spade_norm_fused(x, gb1, rm1, rv1, True, momentum, eps, slope1, groups))


def spade_pair_supported(H, W, up):
    return True


def instnorm_lrelu(x, eps=1e-5, slope=0.2):
    """nn.InstanceNorm2d(affine=False) + LeakyReLU(0.2) (normalization.py:43-44, discriminator.py:374-377)."""
    return F.leaky_relu(F.instance_norm(x, eps=eps), slope)


def boxes_to_layout_single(vecs, boxes, H, W):
    """boxes_to_layout for one frame, exactly the reference's sequence of ops (models/layout.py:28-63,98-130,205-237)."""
    legal = (boxes != 0).any(dim=-1)
    boxes = boxes[legal]
    vecs = vecs[legal]
    O, D = vecs.shape
    b = boxes.view(O, 4, 1, 1)
    x0, y0, ww, hh = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    X = torch.linspace(0, 1, steps=W).view(1, 1, W).to(boxes)
    Y = torch.linspace(0, 1, steps=H).view(1, H, 1).to(boxes)
    X = ((X - x0) / ww).expand(O, H, W)
    Y = ((Y - y0) / hh).expand(O, H, W)
    grid = torch.stack([X, Y], dim=3).mul(2).sub(1)
    img_in = vecs.view(O, D, 1, 1).expand(O, D, 8, 8)
    sampled = F.grid_sample(img_in, grid, align_corners=True)
    return sampled.sum(dim=0, keepdim=True)  # _pool_samples with a single image (scatter_add to row 0)


def layout_render(vecs, boxes, keep, H, W):
    """Batched boxes_to_layout: vecs[BT,O,D], boxes[BT,O,4], keep[BT,O] -> [BT,D,H,W] (generator.py:38-51)."""
    outs = []
    for i in range(vecs.shape[0]):
        m = keep[i].bool()
        if int(m.sum()) == 0:
            outs.append(torch.zeros(1, vecs.shape[2], H, W, dtype=vecs.dtype) + 0 * vecs[i].sum())
        else:
            outs.append(boxes_to_layout_single(vecs[i][m], boxes[i][m], H, W))
    return torch.cat(outs, dim=0)


def layout_conv(U, boxes, keep, H, W, R=3, stride=1, pad=1, residual=None, pool=0):
    """Checker for the low-rank layout convolution: render each object's weight map w_j with the reference's own
    boxes_to_layout arithmetic (models/layout.py:28-63, vectors = identity) and convolve it densely with U as the filter."""
    NB, J, K, _ = U.shape
    outs = []
    eye = torch.eye(J, dtype=U.dtype)
    for n in range(NB):
        wmaps = torch.zeros(1, J, H, W, dtype=U.dtype)
        for j in range(J):
            if bool(keep[n, j]) and bool((boxes[n, j] != 0).any()):
                wmaps[0, j] = boxes_to_layout_single(eye[j:j + 1, j:j + 1], boxes[n, j:j + 1], H, W)[0, 0]
        if pool:
            wmaps = avgpool3s2(wmaps)          # discriminator.py:270-271,351: the second scale sees the pooled input
        filt = U[n].permute(1, 0, 2).reshape(K, J, R, R)
        outs.append(F.conv2d(wmaps, filt, None, stride=stride, padding=pad))
    y = torch.cat(outs, dim=0)
    return y if residual is None else y + residual


def layout_conv_img(U, boxes, keep, H, W, R, pad, img, w_img, gscale=None):
    """Checker for ops.layout_conv_img: the layout channels (layout_conv above) and the image channels (a plain convolution, scaled per
    sample group) of the convolution the reference applies to cat[layouts, frame(s)] (generator.py:66,74,82-83)."""
    return layout_conv(U, boxes, keep, H, W, R, 1, pad, conv2d(img, w_img, None, 1, pad, gscale=gscale), 0)


def layout_conv_img_supported(H, W, R, stride, pool, IC):
    return stride == 1 and pool == 0


def flow_warp(img, flow):
    """resample + get_grid (models/utils.py:113-140), ALIGN_CORNERS=False (data/args.py:228)."""
    b, c, h, w = img.shape
    hor = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, 1, h, w)
    ver = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, 1, h, w)
    grid = torch.cat([hor, ver], 1)
    fl = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
    return F.grid_sample(img, (grid + fl).permute(0, 2, 3, 1), mode='bilinear', padding_mode='border', align_corners=False)


def avgpool3s2(x):
    """MultiscaleActionDiscriminator.downsample (discriminator.py:270-271)."""
    return F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)


def maxpool2(x):
    """nn.MaxPool2d(kernel_size=2, stride=2): the pooling layers of torchvision's vgg19().features (architecture.py:96-110)."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def nearest_resize(x, OH, OW):
    """F.interpolate(..., size, mode='nearest') (normalization.py:102, spade_generator.py:55)."""
    return F.interpolate(x, size=(OH, OW), mode='nearest')


def nearest_pyramid(x, sizes):
    """The per-SPADE F.interpolate(segmap, size, 'nearest') calls of one decoder pass (normalization.py:102), one per size."""
    return tuple(nearest_resize(x, h, w) for (h, w) in sizes)


def upsample2x(x):
    """nn.Upsample(scale_factor=2) (spade_generator.py:36, flows_generator.py:47)."""
    return F.interpolate(x, scale_factor=2, mode='nearest')


def linear(x, w, b=None, relu=False):
    """nn.Linear (+ReLU) of build_mlp (models/layers.py:6-25)."""
    y = F.linear(x, w, b)
    return F.relu(y) if relu else y


def triple_gather(obj, pred, edges):
    """graph.py:64-70."""
    B = obj.shape[0]
    s_idx, o_idx = edges[:, :, 0].contiguous(), edges[:, :, 1].contiguous()
    cs = torch.stack([obj[b, s_idx[b], :] for b in range(B)])
    co = torch.stack([obj[b, o_idx[b], :] for b in range(B)])
    return torch.cat([cs, pred, co], dim=-1)


def triple_pool(newt, edges, valid, O, Hd, Dp):
    """graph.py:73-101 (avg pooling with masked triplets)."""
    B, T = newt.shape[0], newt.shape[1]
    new_s, new_o = newt[:, :, :Hd], newt[:, :, Hd + Dp:]
    s_idx, o_idx = edges[:, :, 0].contiguous(), edges[:, :, 1].contiguous()
    outs = []
    for b in range(B):
        ind = valid[b].bool()
        ss, so = s_idx[b][ind], o_idx[b][ind]
        ns, no = new_s[b][ind], new_o[b][ind]
        pooled = torch.zeros(O, Hd, dtype=newt.dtype)
        pooled = pooled.scatter_add(0, ss.view(-1, 1).expand_as(ns), ns)
        pooled = pooled.scatter_add(0, so.view(-1, 1).expand_as(no), no)
        counts = torch.zeros(O, dtype=newt.dtype)
        ones = torch.ones(T, dtype=newt.dtype)
        counts = counts.scatter_add(0, ss, ones)
        counts = counts.scatter_add(0, so, ones)
        mask = counts > 0
        pooled = torch.where(mask.view(-1, 1), pooled / counts.clamp(min=1).view(-1, 1), pooled)
        outs.append(pooled)
    return torch.stack(outs, dim=0)


def embedding(idx, table):
    """nn.Embedding (attribute_embed.py:38-45)."""
    return F.embedding(idx, table)


def leaky_relu(x, slope=0.2):
    return F.leaky_relu(x, slope)


def tanh(x):
    return torch.tanh(x)


def l1_mean(a, b):
    """nn.L1Loss (loss_model.py:31,39)."""
    return F.l1_loss(a, b)


def hinge_mean(x, kind):
    """GANLoss.loss hinge branch (networks/loss.py:64-76)."""
    if kind == 'g':
        return -torch.mean(x)
    if kind == 'd_real':
        return -torch.mean(torch.min(x - 1, torch.zeros_like(x)))
    return -torch.mean(torch.min(-x - 1, torch.zeros_like(x)))
