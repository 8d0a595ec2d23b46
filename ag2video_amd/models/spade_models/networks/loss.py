"""GAN losses (reference: networks/loss.py:16-98; only the hinge mode is on the hot path, data/args.py:156)."""
import torch.nn as nn

from .... import ops


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
