"""PatchGAN discriminators, reference models/modules/discriminators.py:14-79 (NLayer, `model.{i}` state_dict keys) and :129-226
(SPADENLayer / Multiscale, `discriminator_{i}.model{j}` keys)."""
import functools

import numpy as np
import torch
from torch import nn

from . import nn as cnn
from . import ops
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


class SPADENLayerDiscriminator(BaseNetwork):
    """reference discriminators.py:129-182: `model{i}` children; forward returns every intermediate output."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt):
        super().__init__()
        from .normalization import get_nonspade_norm_layer
        self.opt = opt
        kw = 4
        padw = int(np.ceil((kw - 1.0) / 2))
        nf = opt.ndf
        input_nc = self.compute_D_input_nc(opt)
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        sequence = [[cnn.Conv2d(input_nc, nf, kernel_size=kw, stride=2, padding=padw), cnn.LeakyReLU(0.2, False)]]
        for n in range(1, opt.n_layers_D):
            nf_prev = nf
            nf = min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            sequence += [[norm_layer(cnn.Conv2d(nf_prev, nf, kernel_size=kw, stride=stride, padding=padw)), cnn.LeakyReLU(0.2, False)]]
        sequence += [[cnn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)]]
        for n in range(len(sequence)):
            self.add_module('model' + str(n), cnn.FusedSequential(*sequence[n]))

    def compute_D_input_nc(self, opt):
        return opt.semantic_nc + opt.output_nc

    def forward(self, input):
        results = []
        x = input
        subs = list(self.children())
        for i, submodel in enumerate(subs):
            y = submodel(x)
            if i + 1 < len(subs) and torch.is_grad_enabled() and y.requires_grad:
                tap, x = ops.fanout(y, 2)      # returned to the caller (feature matching) AND fed to the next layer
            else:
                tap = x = y
            results.append(tap)
        return results


class MultiscaleDiscriminator(nn.Module):
    """reference discriminators.py:185-226: num_D PatchGANs on an average-pooled pyramid of the input."""

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--num_D', type=int, default=2, help='number of discriminators to be used in multiscale')
        parser.add_argument('--norm_D', type=str, default='spectralinstance', help='instance normalization or batch normalization')
        parser.set_defaults(n_layers_D=4)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, SPADENLayerDiscriminator(opt))

    def downsample(self, input):
        return ops.AvgPool3x3s2Fn.apply(input)

    def forward(self, input):
        result = []
        nd = len(self._modules)
        for i, (name, D) in enumerate(self.named_children()):
            if i + 1 < nd and input.requires_grad and torch.is_grad_enabled():
                cur, nxt = ops.fanout(input, 2)
            else:
                cur = nxt = input
            result.append(D(cur))
            if i + 1 < nd:              # the reference also pools after the last scale; the result is unused
                input = self.downsample(nxt)
        return result
