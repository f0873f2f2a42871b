"""InceptionSPADEGenerator (GauGAN student / teacher): same constructor, attribute names, state_dict keys and forward contract
(`mapping_layers` -> (image, {name: activation})) as the reference's
models/modules/inception_architecture/inception_spade_generator.py:14-143, every op a gfx950 kernel."""
import torch
from torch import nn

from . import nn as cnn
from . import ops
from .inception_generator import BaseNetwork
from .inception_modules import SPADEInvertedResidualChannels, _get_named_block_list, seg_at


class InceptionSPADEGenerator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt):
        super(InceptionSPADEGenerator, self).__init__()
        self.opt = opt
        nf = opt.ngf
        self.fc_norm = cnn.SynchronizedBatchNorm2d(16 * nf, affine=True)
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        self.fc = cnn.Conv2d(self.opt.semantic_nc, 16 * nf, 3, padding=1)
        self.head_0 = SPADEInvertedResidualChannels(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEInvertedResidualChannels(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEInvertedResidualChannels(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEInvertedResidualChannels(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEInvertedResidualChannels(8 * nf, 4 * nf, opt)
        self.up_2 = SPADEInvertedResidualChannels(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEInvertedResidualChannels(2 * nf, 1 * nf, opt)
        final_nc = nf
        if opt.num_upsampling_layers == 'most':
            self.up_4 = SPADEInvertedResidualChannels(1 * nf, nf // 2, opt)
            final_nc = nf // 2
        self.conv_img = cnn.Conv2d(final_nc, 3, 3, padding=1)
        self.up = cnn.Upsample(scale_factor=2)
        self._lrelu = cnn.LeakyReLU(2e-1)
        self._tanh = cnn.Tanh()

    def compute_latent_vector_size(self, opt):
        if opt.num_upsampling_layers == 'normal':
            num_up_layers = 5
        elif opt.num_upsampling_layers == 'more':
            num_up_layers = 6
        elif opt.num_upsampling_layers == 'most':
            num_up_layers = 7
        else:
            raise ValueError('opt.num_upsampling_layers [%s] not recognized' % opt.num_upsampling_layers)
        sw = opt.crop_size // (2 ** num_up_layers)
        sh = round(sw / opt.aspect_ratio)
        return sw, sh

    def forward(self, input, mapping_layers=[]):
        seg = ops.conform(input)
        ret_acts = {}

        def keep(name, t):
            """Tap `t` for the distillation loss; returns the tensor the network continues with (two consumers -> FanoutFn)."""
            if name not in mapping_layers:
                return t
            if torch.is_grad_enabled() and t.requires_grad:
                ret_acts[name], t = ops.fanout(t, 2)
            else:
                ret_acts[name] = t
            return t

        x = seg_at(seg, (self.sh, self.sw))
        x = keep('fc', self.fc_norm(self.fc(x)))
        x = keep('head_0', self.head_0(x, seg))
        x = self.up(x)
        x = keep('G_middle_0', self.G_middle_0(x, seg))
        if self.opt.num_upsampling_layers in ('more', 'most'):
            x = self.up(x)
        x = keep('G_middle_1', self.G_middle_1(x, seg))
        for name in ('up_0', 'up_1', 'up_2', 'up_3'):
            x = self.up(x)
            x = keep(name, getattr(self, name)(x, seg))
        if self.opt.num_upsampling_layers == 'most':
            x = self.up(x)
            x = keep('up_4', self.up_4(x, seg))
        x = self.conv_img(self._lrelu(x), fuse_act=self._tanh)      # F.leaky_relu(x, 2e-1) -> conv_img -> tanh (:117-118)
        if len(mapping_layers) == 0:
            return x
        return x, ret_acts

    def remove_spectral_norm(self):
        raise NotImplementedError('remove_spectral_norm belongs to the export path (out of scope)')

    def get_named_block_list(self):
        return _get_named_block_list(self, spade=True, num_upsampling_layers=self.opt.num_upsampling_layers)
