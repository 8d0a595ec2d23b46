"""GAN losses (reference: networks/loss.py:16-98; only the hinge mode is on the hot path, data/args.py:156) and the VGG
perceptual loss (loss.py:102-118)."""
import torch
import torch.nn as nn

from .... import ops
from .architecture import VGG19


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=None, opt=None):
        super().__init__()
        if gan_mode != 'hinge':
            raise ValueError('Unexpected gan_mode {} (only hinge is on the hot path)'.format(gan_mode))
        self.gan_mode = gan_mode
        self.opt = opt

    def loss(self, x, target_is_real, for_discriminator=True):
        if for_discriminator:
            return ops.hinge_mean(x, 'd_real' if target_is_real else 'd_fake')
        assert target_is_real, "The generator's hinge loss must be aiming for real"
        return ops.hinge_mean(x, 'g')

    def __call__(self, x, target_is_real, for_discriminator=True):
        """Multiscale input (list of per-scale feature lists): last map of each scale, averaged (loss.py:87-96);
        returns a 1-element tensor as the reference does."""
        if isinstance(x, list):
            total = 0
            for pred in x:
                if isinstance(pred, list):
                    pred = pred[-1]
                total = total + self.loss(pred, target_is_real, for_discriminator).view(1)
            return total / len(x)
        return self.loss(x, target_is_real, for_discriminator)


class VGGLoss(nn.Module):
    """Perceptual loss (reference: networks/loss.py:102-118): sum_i w_i * L1(vgg_i(x), vgg_i(y).detach()) over the five VGG19
    slices, w = 1/32, 1/16, 1/8, 1/4, 1; x, y are [B, T, 3, H, W].  The target branch runs without a graph."""

    def __init__(self, gpu_ids=None):
        super().__init__()
        self.vgg = VGG19()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        bs, nt, ch, h, w = x.size()
        x = x.contiguous().view(-1, ch, h, w)
        y = y.contiguous().view(-1, ch, h, w)
        with ops.bf16x3_min_rows(48):
            x_vgg = self.vgg(x)
            with torch.no_grad():
                y_vgg = self.vgg(y)
        loss = 0
        for i in range(len(x_vgg)):
            loss = loss + self.weights[i] * ops.l1_mean(x_vgg[i], y_vgg[i].detach())
        return loss
