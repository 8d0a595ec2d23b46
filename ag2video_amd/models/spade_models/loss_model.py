"""Loss facade (reference: models/spade_models/loss_model.py:14-149)."""
import contextlib
import os

import torch
import torch.nn.functional as F

from ... import ops
from ..utils import resample
from .networks.loss import GANLoss, VGGLoss


@contextlib.contextmanager
def _params_frozen(module):
    """The generator step back-propagates *through* the discriminator; the gradients it would deposit in the
    discriminator's own parameters are discarded by optimizer_d_img.zero_grad() before they are ever read
    (scripts/train.py:457-459,522-524), so their weight-gradient kernels are skipped altogether."""
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


class LossModel(torch.nn.Module):
    def __init__(self, opt, discriminator):
        super().__init__()
        self.opt = opt
        self.n_frames_G = opt.n_frames_G
        self.n_frames_D = opt.n_frames_D
        self.discriminator = discriminator
        if hasattr(discriminator, 'img_discriminator'):
            self.netD_img = discriminator.img_discriminator
        self.netD_tmp = getattr(discriminator, 'temporal_discriminator', None)       # --temporal_D 1 (meta_models.py)
        self.criterionGAN = GANLoss(opt.gan_mode, opt=opt)
        if not opt.no_vgg_loss:
            # loss_model.py:35-36.  The reference downloads ImageNet weights through torchvision; there is no network here: the
            # weights are PyTorch's default init until the caller loads some (--vgg_weights: a torchvision vgg19 state dict file;
            # engine.TrainStep(weights_seed=...) fills them closed-form like every other parameter).
            path = getattr(opt, 'vgg_weights', None)
            if not path and not getattr(opt, 'vgg_random_init', False):
                # the reference's default objective includes lambda_vgg * VGG on ImageNet features; without those weights the term is
                # noise with weight 10 in total_loss -- never silently
                raise RuntimeError("perceptual loss: no ImageNet VGG19 weights (the reference downloads them; no network here). Pass "
                                   "--vgg_weights <torchvision vgg19 state dict>, or --no_vgg_loss, or opt in to a randomly "
                                   "initialised VGG19 with --vgg_random_init (throughput / parity runs only)")
            self.criterionVGG = VGGLoss(opt.gpu_ids)
            self.vgg_weights_loaded = bool(path)
            if path:
                sd = torch.load(path, map_location='cpu')
                self.criterionVGG.vgg.load_torchvision_features({k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.')})

    # ---- discriminator layout terms, once per step.  The terms depend on the discriminator's parameters and on the layout inputs
    # (objects, ground-truth boxes, action data) only.  The generator step and the discriminator step of one training iteration
    # see the same inputs and -- optimizer_d_img steps after the discriminator loss -- the same discriminator parameters, so the
    # reference evaluates the same function twice (four times counting fake/real, loss_model.py:73,77,118,120).  Here the
    # generator loss computes it with its autograd graph (outside _params_frozen), uses it detached (nothing in it depends on
    # the generator), and leaves it for the discriminator loss of the same batch, which back-propagates through it.  The key
    # holds the input tensors themselves (storage + version) and every discriminator parameter's version counter: any in-place
    # change in between, another batch, or a graph net with batch statistics falls back to a fresh evaluation.
    def _terms_key(self, inputs):
        # FusedAdam writes the parameters through raw pointers (no version bump): its step counter is part of the key
        opt_d = getattr(self.discriminator, 'optimizer_d_img', None)
        return (tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())) for t in inputs),
                tuple(p._version for p in self.netD_img.parameters()), id(opt_d), getattr(opt_d, '_step', None))

    def _layout_terms(self, objs, r_boxes, r_adata, producer, batch=None):
        """producer=True (generator loss): evaluate and remember; producer=False (discriminator loss): reuse when valid."""
        if not hasattr(self.netD_img, 'layout_terms'):
            return None
        inputs = [objs, r_boxes] + list(r_adata)
        shareable = (torch.is_grad_enabled() and getattr(self.opt, 'mlp_normalization', 'none') == 'none'
                     and not any(t.requires_grad for t in inputs if t.is_floating_point())
                     and not getattr(self.opt, 'no_shared_layout_terms', False))
        cached, self._terms_cache = getattr(self, '_terms_cache', None), None
        pre, self._terms_prefetch = getattr(self, '_terms_prefetch', None), None
        if not shareable:
            return self.netD_img.layout_terms(objs, r_boxes, r_adata)
        key = self._terms_key(inputs)
        if producer and pre is not None and batch is not None and pre[0] == self._prefetch_key(batch):
            torch.cuda.current_stream().wait_event(pre[2])
            self._terms_cache = (key, inputs, pre[1])
            return pre[1]
        if not producer and cached is not None and cached[0] == key:
            return cached[2]
        terms = self.netD_img.layout_terms(objs, r_boxes, r_adata)
        if producer and terms is not None:
            self._terms_cache = (key, inputs, terms)
        return terms

    def prefetch_layout_terms(self, batch, actions_data, stream, after=None):
        """Evaluate the discriminator's layout terms for `batch` on `stream` (engine.TrainStep: a side stream, before the generator
        forward's launches are queued).  They depend on D's parameters and the batch only, so they can be computed while the generator runs; the
        generator loss picks them up (after waiting for `stream`) when it is called with the same batch tensors and unchanged
        parameters, and installs them under the regular key for the discriminator loss."""
        self._terms_prefetch = None
        if not hasattr(self.netD_img, 'layout_terms') or not torch.is_grad_enabled():
            return
        imgs, objs, boxes, triplets, actions, _ = batch
        nc = self.n_frames_G - 1
        r_boxes = boxes[:, nc:]
        r_adata = [item[:, nc:] for item in actions_data]
        inputs = [objs, r_boxes] + list(r_adata)
        if (getattr(self.opt, 'mlp_normalization', 'none') != 'none' or getattr(self.opt, 'no_shared_layout_terms', False)
                or any(t.requires_grad for t in inputs if t.is_floating_point())):
            return
        main = torch.cuda.current_stream()
        if after is not None:       # an event on the caller's stream after which batch and parameters are final
            stream.wait_event(after)
        else:
            stream.wait_stream(main)
        with torch.cuda.stream(stream):
            terms = self.netD_img.layout_terms(objs, r_boxes, r_adata)
            if terms is None:
                return
            for t in terms:
                t.record_stream(main)
            done = torch.cuda.Event()
            done.record(stream)
        self._terms_prefetch = (self._prefetch_key(batch), terms, done)

    def _prefetch_key(self, batch):
        imgs, objs, boxes, triplets, actions, _ = batch
        return self._terms_key([objs, boxes, triplets, actions])

    def compute_graph_loss(self, batch, boxes_pred):
        """Masked smooth-L1 on predicted boxes (loss_model.py:41-60); a few thousand elements: host-side torch ops."""
        imgs, objs, boxes, triplets, actions, video_id = batch
        bp = boxes_pred[:, 1:].contiguous().view(-1, 4)
        bg = boxes[:, 1:].contiguous().view(-1, 4)
        flat = F.smooth_l1_loss(bp, bg, reduction='none') * self.opt.bbox_pred_loss_weight
        fo = objs.unsqueeze(1).repeat(1, boxes.size(1) - 1, 1, 1).view(-1, objs.size(-1))
        if objs.size(-1) > 1:
            mask = (fo.sum(1) != 0).repeat(4, 1).permute(1, 0)
        else:
            mask = (fo != 0)
        G = {"bbox_pred": (flat * mask.to(flat.dtype)).mean()}
        G['total_loss'] = torch.stack(list(G.values()), dim=0).sum()
        return G

    def _temporal_windows(self, x):
        """[B, T, 3, H, W] -> [B * (T // fpa), fpa, 3, H, W]: non-overlapping windows of frames_per_action consecutive frames (the
        temporal discriminator's input is frames_per_action * 3 channels, discriminator.py:46-47; a remainder is dropped)."""
        fpa = self.opt.frames_per_action
        B, T = x.shape[0], x.shape[1]
        nw = T // fpa
        if nw < 1:
            raise RuntimeError("--temporal_D 1 needs at least frames_per_action = %d frames after the first n_frames_G - 1; got %d" % (fpa, T))
        return x[:, :nw * fpa].reshape((B * nw, fpa) + tuple(x.shape[2:]))

    def compute_generator_loss(self, batch, model_out):
        imgs, objs, boxes, triplets, actions, video_id = batch
        imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data = model_out
        nc = self.n_frames_G - 1
        r_imgs, r_boxes, r_pred = imgs[:, nc:], boxes[:, nc:], imgs_pred[:, nc:]
        r_adata = [item[:, nc:] for item in actions_data]
        G = {}
        # the layout part of D's first convolutions is the same for the fake and the real pass (and for the discriminator
        # loss of this batch: _layout_terms)
        terms = self._layout_terms(objs, r_boxes, r_adata, producer=True, batch=batch)
        if terms is not None:
            terms = [t.detach() for t in terms]
        with _params_frozen(self.netD_img):
            fake = self.netD_img(r_pred, objs, r_boxes, r_adata, **({'layout_terms': terms} if terms is not None else {}))
        G['GAN_Img'] = self.criterionGAN(fake, True, for_discriminator=False).squeeze(0) * \
            self.opt.discriminator_img_loss_weight
        if not self.opt.no_ganFeat_loss:
            with torch.no_grad():           # the real features are only used detached (loss_model.py:84)
                real = self.netD_img(r_imgs, objs, r_boxes, r_adata, **({'layout_terms': terms} if terms is not None else {}))
            num_D = len(fake)
            feat = torch.zeros(1, device=imgs.device, dtype=torch.float32)
            for i in range(num_D):
                for j in range(len(fake[i]) - 1):
                    feat = feat + ops.l1_mean(fake[i][j], real[i][j].detach()) * self.opt.lambda_feat / num_D
            G['GAN_Feat'] = feat.squeeze(0)
        if self.netD_tmp is not None:       # temporal PatchGAN over frame windows: the image discriminator's two terms once more
            with _params_frozen(self.netD_tmp):
                fake_t = self.netD_tmp(self._temporal_windows(r_pred))
            G['GAN_Tmp'] = self.criterionGAN(fake_t, True, for_discriminator=False).squeeze(0) * self.opt.discriminator_img_loss_weight
            if not self.opt.no_ganFeat_loss:
                with torch.no_grad():
                    real_t = self.netD_tmp(self._temporal_windows(r_imgs))
                feat = torch.zeros(1, device=imgs.device, dtype=torch.float32)
                for i in range(len(fake_t)):
                    for j in range(len(fake_t[i]) - 1):
                        feat = feat + ops.l1_mean(fake_t[i][j], real_t[i][j].detach()) * self.opt.lambda_feat / len(fake_t)
                G['GAN_Tmp_Feat'] = feat.squeeze(0)
        b, t, c, h, w = imgs.size()
        prev = imgs[:, nc - 1:-1].contiguous().view(-1, c, h, w)
        nxt = imgs[:, nc:].contiguous().view(-1, c, h, w)
        fl = flows_pred[:, nc - 1:-1].contiguous().view(-1, 2, h, w)
        G["loss_F_Warp"] = ops.l1_mean(resample(prev, fl), nxt) * self.opt.lambda_F_warp
        if not self.opt.no_vgg_loss:
            G['VGG'] = self.criterionVGG(r_pred, r_imgs) * self.opt.lambda_vgg               # loss_model.py:99-100
        G['total_loss'] = torch.stack(list(G.values()), dim=0).sum()
        return G

    def compute_discriminator_loss(self, batch, model_out):
        imgs, objs, boxes, triplets, actions, _ = batch
        imgs_pred, boxes_pred, flows_pred, conf_pred, actions_data = model_out
        nc = self.n_frames_G - 1
        r_adata = [item[:, nc:] for item in actions_data]
        terms = self._layout_terms(objs, boxes[:, nc:], r_adata, producer=False)
        kw = {'layout_terms': terms} if terms is not None else {}
        fake = self.netD_img(imgs_pred[:, nc:].detach(), objs, boxes[:, nc:], r_adata, **kw)
        real = self.netD_img(imgs[:, nc:], objs, boxes[:, nc:], r_adata, **kw)
        D = {"D_img_fake": self.criterionGAN(fake, False, for_discriminator=True),
             "D_img_real": self.criterionGAN(real, True, for_discriminator=True)}
        if self.netD_tmp is not None:
            D["D_tmp_fake"] = self.criterionGAN(self.netD_tmp(self._temporal_windows(imgs_pred[:, nc:].detach())), False, for_discriminator=True)
            D["D_tmp_real"] = self.criterionGAN(self.netD_tmp(self._temporal_windows(imgs[:, nc:])), True, for_discriminator=True)
        D["total_img_loss"] = torch.stack(list(D.values()), dim=0).sum()
        return D

    def use_gpu(self):
        return len(self.opt.gpu_ids) > 0

    def forward(self, batch, model_out, mode):
        if mode == "compute_discriminator_loss":
            return self.compute_discriminator_loss(batch, model_out)
        if mode == "compute_generator_loss":
            return self.compute_generator_loss(batch, model_out)
        if mode == "compute_graph_loss":
            return self.compute_graph_loss(batch, model_out)
        raise ValueError("unknown LossModel mode %r" % (mode,))
