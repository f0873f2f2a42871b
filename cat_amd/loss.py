"""Losses of the distillation step: GANLoss (reference models/modules/loss.py:8-99: hinge, lsgan, vanilla, wgangp), the L1 / MSE
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
        if gan_mode not in ('lsgan', 'hinge', 'vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        self.gan_mode = gan_mode

    def __call__(self, prediction, target_is_real, for_discriminator=True):
        if self.gan_mode == 'lsgan':
            target = self.real_label if target_is_real else self.fake_label
            return ops.LossFn.apply(prediction, None, L.LOSS_LSGAN, target)
        if self.gan_mode == 'vanilla':      # nn.BCEWithLogitsLoss against the expanded label, loss.py:33-34,63-65
            target = self.real_label if target_is_real else self.fake_label
            return ops.LossFn.apply(prediction, None, L.LOSS_BCE_LOGITS, target)
        if self.gan_mode == 'wgangp':       # loss.py:66-70
            return ops.LossFn.apply(prediction, None, L.LOSS_NEG_MEAN if target_is_real else L.LOSS_MEAN, 0.0)
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


class VGG19(nn.Module):
    """torchvision vgg19().features[0:30] cut into the reference's five slices (models/modules/loss.py:151-186), same
    `slice{k}.{idx}.weight` state_dict keys, every layer a gfx950 kernel (3x3 conv + fused ReLU, 2x2 max-pool).

    The reference downloads torchvision's ImageNet weights; there is no network here, so the constructor leaves torch's default
    initialisation in place and `load_torchvision_state_dict` / `load_state_dict` install real weights.
    `width_div` > 1 builds a proportionally narrower net (tests)."""
    CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
    SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]

    def __init__(self, requires_grad=False, width_div=1):
        super().__init__()
        from . import nn as cnn
        layers, cin = [], 3
        for v in self.CFG:
            if v == 'M':
                layers.append(cnn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [cnn.Conv2d(cin, v // width_div, kernel_size=3, padding=1), cnn.ReLU(inplace=True)]
                cin = v // width_div
        for k, (a, b) in enumerate(self.SLICES):
            seq = cnn.FusedSequential()
            for x in range(a, b):
                seq.add_module(str(x), layers[x])
            setattr(self, 'slice%d' % (k + 1), seq)
        if not requires_grad:
            for param in self.parameters():
                param.requires_grad = False

    def load_torchvision_state_dict(self, sd):
        """`sd`: torchvision.models.vgg19().state_dict() (keys 'features.{idx}.weight') or its `.features` part ('{idx}.weight')."""
        own = self.state_dict()
        for k in own:
            idx_key = k.split('.', 1)[1]
            src = sd.get('features.' + idx_key, sd.get(idx_key))
            if src is None:
                raise KeyError('VGG19 weights: missing ' + idx_key)
            own[k].copy_(src)

    def forward(self, X):
        h1 = self.slice1(X)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]


class VGGLoss(nn.Module):
    """models/modules/loss.py:189-203: sum_i w_i * L1(vgg(x)_i, vgg(y)_i.detach())."""

    def __init__(self, width_div=1):
        super(VGGLoss, self).__init__()
        self.vgg = VGG19(width_div=width_div)
        self.vgg.eval()
        self.criterion = L1Loss()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def terms(self, x, y):
        """[(w_i, L1_i)]: the caller seeds the backward with the weights (no torch arithmetic on the hot path)."""
        import torch
        grad = torch.is_grad_enabled() and x.requires_grad
        feats = []
        h = x
        for k in range(5):
            h = getattr(self.vgg, 'slice%d' % (k + 1))(h)
            if grad and k < 4:
                keep, h = ops.fanout(h, 2)      # tapped by the loss AND consumed by the next slice
            else:
                keep = h
            feats.append(keep)
        with torch.no_grad():
            y_vgg = self.vgg(y)
        return [(self.weights[i], self.criterion(feats[i], y_vgg[i])) for i in range(5)]

    def forward(self, x, y):
        loss = 0
        for w, t in self.terms(x, y):
            loss = loss + w * t
        return loss
