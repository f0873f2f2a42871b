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

    # (attribute, width multiplier in, width multiplier out) of the SPADE blocks, in registration (= checkpoint key) order
    _BLOCKS = (('head_0', 16, 16), ('G_middle_0', 16, 16), ('G_middle_1', 16, 16), ('up_0', 16, 8), ('up_1', 8, 4), ('up_2', 4, 2),
               ('up_3', 2, 1))
    _UP_LAYERS = {'normal': 5, 'more': 6, 'most': 7}      # x2 upsamplings between the latent map and the output

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        base = opt.ngf
        self.fc_norm = cnn.SynchronizedBatchNorm2d(16 * base, affine=True)
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        self.fc = cnn.Conv2d(opt.semantic_nc, 16 * base, 3, padding=1)
        for attr, m_in, m_out in self._BLOCKS:
            setattr(self, attr, SPADEInvertedResidualChannels(m_in * base, m_out * base, opt))
        out_width = base
        if opt.num_upsampling_layers == 'most':
            out_width = base // 2
            self.up_4 = SPADEInvertedResidualChannels(base, out_width, opt)
        self.conv_img = cnn.Conv2d(out_width, 3, 3, padding=1)
        self.up = cnn.Upsample(scale_factor=2)
        self._lrelu = cnn.LeakyReLU(2e-1)
        self._tanh = cnn.Tanh()

    def compute_latent_vector_size(self, opt):
        """(width, height) of the map the generator starts from: crop_size / 2^(number of upsamplings), height by the aspect ratio."""
        if opt.num_upsampling_layers not in self._UP_LAYERS:
            raise ValueError('opt.num_upsampling_layers [%s] not recognized' % opt.num_upsampling_layers)
        width = opt.crop_size // (1 << self._UP_LAYERS[opt.num_upsampling_layers])
        return width, round(width / opt.aspect_ratio)

    def forward(self, input, mapping_layers=[]):
        if self.training:
            # one operand-preparation launch per pass for ALL fused units: their plans are grouped once they exist (second forward) and again
            # whenever a plan was rebuilt since (fused_spade.PLAN_GEN moves: a reducer's broadcast drops the plans, units become applicable
            # later) -- a plan without a group would silently fall back to one launch per unit.  The plans hold a weak reference to this module.
            from . import fused_block, fused_spade
            if self.__dict__.get('_cat_prep_gen') != fused_spade.PLAN_GEN:
                units = fused_spade.units_of(self)
                if units:
                    import weakref
                    fused_block.prepare_plans(units, weakref.ref(self), False)
                self.__dict__['_cat_prep_gen'] = fused_spade.PLAN_GEN      # also without units: the module walk is repeated only when PLAN_GEN moves
        seg = ops.conform(input)
        ret_acts = {}
        self._gb_prepass(seg)

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
        return (x, ret_acts) if len(mapping_layers) else x

    def _block_sizes(self):
        """(block name, spatial size its SPADE layer runs at), in forward order (the up-sampling schedule of forward())."""
        h, w = self.sh, self.sw
        out = [('head_0', (h, w))]
        h, w = 2 * h, 2 * w
        out.append(('G_middle_0', (h, w)))
        if self.opt.num_upsampling_layers in ('more', 'most'):
            h, w = 2 * h, 2 * w
        out.append(('G_middle_1', (h, w)))
        for name in ('up_0', 'up_1', 'up_2', 'up_3'):
            h, w = 2 * h, 2 * w
            out.append((name, (h, w)))
        if self.opt.num_upsampling_layers == 'most':
            out.append(('up_4', (2 * h, 2 * w)))
        return out

    def _gb_prepass(self, seg):
        """Under SynchronizedBatchNorm over several ranks: the gamma|beta nets of ALL SPADE layers read only the segmentation map (reference
        inception_modules.py:746-762), so they run first, in lockstep, with their statistics exchanges merged -- two collectives per pass
        instead of two per layer (fused_spade.prepass); every InceptionSPADE then finds its [gamma | beta] ready.  A no-op on one rank."""
        # results of an earlier pre-pass that no SPADE layer consumed (a forward that aborted midway) must never meet another input: dropped
        # on EVERY entry, also when this pass will not run a pre-pass (eval mode, reducer gone)
        for name, _ in self._block_sizes():
            getattr(self, name).spade.__dict__.pop('_cat_gb_pre', None)
        if not self.training or ops.bn_sync() is None:
            return
        from . import fused_spade
        units, owners = [], []
        for name, size in self._block_sizes():
            blk = getattr(self, name)
            sp = blk.spade
            if len(sp.res_ops) + len(sp.dw_ops) == 0 or len(blk.res_ops) + len(blk.dw_ops) == 0:
                continue      # no gamma|beta net, or a block pruned down to its shortcut (its SPADE layer never runs)
            units.append((sp, sp.res_ops, sp.dw_ops, sp.input_dim, 2 * sp.output_dim, seg_at(seg, size)))
            owners.append(sp)
        gbs = fused_spade.prepass(units)
        if gbs is not None:
            for sp, gb in zip(owners, gbs):
                sp.__dict__['_cat_gb_pre'] = (gb, seg.data_ptr(), seg._version)      # tagged with the map it was computed from

    def remove_spectral_norm(self):
        """Reference inception_spade_generator.py:126-137 (export path)."""
        names = ['head_0', 'G_middle_0', 'G_middle_1', 'up_0', 'up_1', 'up_2', 'up_3']
        if self.opt.num_upsampling_layers == 'most':
            names.append('up_4')
        for name in names:
            getattr(self, name).remove_spectral_norm()

    def get_named_block_list(self):
        return _get_named_block_list(self, spade=True, num_upsampling_layers=self.opt.num_upsampling_layers)
