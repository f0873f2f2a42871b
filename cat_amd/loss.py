"""Losses of the distillation step: GANLoss (reference models/modules/loss.py:8-99, hinge + lsgan), the L1 / MSE
reconstruction criteria and KA (utils/common.py:38-46).  Each evaluates to a 0-d device tensor produced (and
differentiated) by the reduction kernels of libcat_hip.so."""
from torch import nn

from . import _lib as L
from . import ops


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super(GANLoss, self).__init__()
        self.real_label = float(target_real_label)
        self.fake_label = float(target_fake_label)
        if gan_mode not in ('lsgan', 'hinge'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        self.gan_mode = gan_mode

    def __call__(self, prediction, target_is_real, for_discriminator=True):
        if self.gan_mode == 'lsgan':
            target = self.real_label if target_is_real else self.fake_label
            return ops.LossFn.apply(prediction, None, L.LOSS_LSGAN, target)
        if isinstance(prediction, list):   # multiscale form, loss.py:71-82
            loss = 0
            for pred_i in prediction:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss = loss + self(pred_i, target_is_real, for_discriminator)
            return loss / len(prediction)
        if for_discriminator:
            kind = L.LOSS_HINGE_D_REAL if target_is_real else L.LOSS_HINGE_D_FAKE
            return ops.LossFn.apply(prediction, None, kind, 0.0)
        assert target_is_real
        return ops.LossFn.apply(prediction, None, L.LOSS_NEG_MEAN, 0.0)


class L1Loss(nn.Module):
    def forward(self, input, target):
        return ops.LossFn.apply(input, target, L.LOSS_L1, 0.0)


class MSELoss(nn.Module):
    def forward(self, input, target):
        return ops.LossFn.apply(input, target, L.LOSS_MSE, 0.0)


def KA(X, Y):
    return ops.ka(X, Y)
