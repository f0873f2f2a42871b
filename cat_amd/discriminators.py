"""PatchGAN discriminator, reference models/modules/discriminators.py:14-79 (same `model.{i}` state_dict keys)."""
import functools

from torch import nn

from . import nn as cnn
from .inception_generator import BaseNetwork
from .inception_modules import get_active_fn


class NLayerDiscriminator(BaseNetwork):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=cnn.BatchNorm2d, active_fn='nn.LeakyReLU'):
        super(NLayerDiscriminator, self).__init__()
        if type(norm_layer) == functools.partial:
            use_bias = issubclass(norm_layer.func, nn.InstanceNorm2d)
        else:
            use_bias = issubclass(norm_layer, nn.InstanceNorm2d)
        active_fn = get_active_fn(active_fn)
        kw, padw = 4, 1
        sequence = [cnn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), active_fn(0.2)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev = nf_mult
            nf_mult = min(2 ** n, 8)
            sequence += [cnn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), active_fn(0.2)]
        nf_mult_prev = nf_mult
        nf_mult = min(2 ** n_layers, 8)
        sequence += [cnn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), active_fn(0.2)]
        sequence += [cnn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = cnn.FusedSequential(*sequence)

    def forward(self, input):
        return self.model(input)
