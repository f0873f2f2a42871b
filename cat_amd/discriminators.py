"""PatchGAN discriminators, reference models/modules/discriminators.py:14-79 (NLayer, `model.{i}` state_dict keys) and :129-226
(SPADENLayer / Multiscale, `discriminator_{i}.model{j}` keys)."""
import functools

import torch
from torch import nn

from . import nn as cnn
from . import ops
from .inception_generator import BaseNetwork
from .inception_modules import get_active_fn


class NLayerDiscriminator(BaseNetwork):
    """70x70-style PatchGAN: 4x4 convs, stride 2 for the first `n_layers` (widths ndf * min(2^i, 8)), then two stride-1 convs down to one
    channel; every conv but the first and the last is followed by norm + LeakyReLU(0.2) and carries a bias only with InstanceNorm."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=cnn.BatchNorm2d, active_fn='nn.LeakyReLU'):
        super().__init__()
        norm_cls = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        bias = issubclass(norm_cls, nn.InstanceNorm2d)
        act = get_active_fn(active_fn)
        widths = [ndf * min(2 ** i, 8) for i in range(n_layers + 1)]
        layers = [cnn.Conv2d(input_nc, widths[0], kernel_size=4, stride=2, padding=1), act(0.2)]
        for i in range(1, n_layers + 1):
            stride = 2 if i < n_layers else 1
            layers += [cnn.Conv2d(widths[i - 1], widths[i], kernel_size=4, stride=stride, padding=1, bias=bias), norm_layer(widths[i]), act(0.2)]
        layers.append(cnn.Conv2d(widths[-1], 1, kernel_size=4, stride=1, padding=1))
        self.model = cnn.FusedSequential(*layers)

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
        wrap = get_nonspade_norm_layer(opt, opt.norm_D)
        pad = 2                                              # ceil((4 - 1) / 2)
        widths = [opt.ndf]
        for _ in range(1, opt.n_layers_D):
            widths.append(min(2 * widths[-1], 512))
        stages = [[cnn.Conv2d(self.compute_D_input_nc(opt), widths[0], kernel_size=4, stride=2, padding=pad), cnn.LeakyReLU(0.2, False)]]
        for i in range(1, opt.n_layers_D):
            stride = 2 if i < opt.n_layers_D - 1 else 1
            stages.append([wrap(cnn.Conv2d(widths[i - 1], widths[i], kernel_size=4, stride=stride, padding=pad)), cnn.LeakyReLU(0.2, False)])
        stages.append([cnn.Conv2d(widths[-1], 1, kernel_size=4, stride=1, padding=pad)])
        for i, stage in enumerate(stages):
            self.add_module('model%d' % i, cnn.FusedSequential(*stage))

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
