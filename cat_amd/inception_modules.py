"""Student / teacher building blocks: same classes, constructor arguments, attribute names and state_dict keys as
the reference's models/modules/inception_modules.py:12-243 (get_active_fn, ConvBNReLU, InvertedResidualChannels),
built from cat_amd.nn layers so every op is a gfx950 kernel."""
import collections
import functools
import re

import torch
from torch import nn

from . import frozen
from . import fused_block
from . import nn as cnn
from . import ops


def add_prefix(name, prefix=None, split='.'):
    """reference common.py add_prefix."""
    if prefix is not None:
        return '{}{}{}'.format(prefix, split, name)
    return name


def get_active_fn(name):
    """reference inception_modules.py:12-19."""
    active_fn = {
        'nn.ReLU6': functools.partial(cnn.ReLU6, inplace=True),
        'nn.ReLU': functools.partial(cnn.ReLU, inplace=True),
        'nn.LeakyReLU': functools.partial(cnn.LeakyReLU, inplace=True),
    }[name]
    return active_fn


class ConvBNReLU(cnn.FusedSequential):
    """Conv2d(pad 0) -> norm -> activation, reference inception_modules.py:22-44."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1, use_bias=True,
                 norm_layer=cnn.InstanceNorm2d, norm_kwargs=None, active_fn=None):
        if norm_kwargs is None:
            norm_kwargs = {}
        super(ConvBNReLU, self).__init__(
            cnn.Conv2d(in_planes, out_planes, kernel_size, stride, 0, groups=groups, bias=use_bias),
            norm_layer(out_planes, **norm_kwargs), active_fn())


class InvertedResidualChannels(nn.Module):
    """x + pw_bn(sum_k res_k(x) + sum_k dw_k(x)), reference inception_modules.py:47-243."""

    def __init__(self, inp, res_channels, dw_channels, channels_reduction_factor, res_kernel_sizes, dw_kernel_sizes,
                 padding_type='reflect', use_bias=True, norm_layer=cnn.InstanceNorm2d, norm_kwargs=None, dropout_rate=0.0,
                 active_fn=None):
        super(InvertedResidualChannels, self).__init__()
        if type(res_kernel_sizes) == int:
            res_kernel_sizes = [res_kernel_sizes]
        if res_channels is not None:
            assert type(res_channels) == int or len(res_channels) == len(res_kernel_sizes)
        if type(dw_kernel_sizes) == int:
            dw_kernel_sizes = [dw_kernel_sizes]
        if dw_channels is not None:
            assert type(dw_channels) == int or len(dw_channels) == len(dw_kernel_sizes)

        self.input_dim = inp
        if res_channels is None:
            self.res_channels = [inp // channels_reduction_factor for _ in res_kernel_sizes]
        elif type(res_channels) == int:
            self.res_channels = [res_channels // channels_reduction_factor for _ in res_kernel_sizes]
        else:
            self.res_channels = [c // channels_reduction_factor for c in res_channels]
        if dw_channels is None:
            self.dw_channels = [inp // channels_reduction_factor for _ in dw_kernel_sizes]
        elif type(dw_channels) == int:
            self.dw_channels = [dw_channels // channels_reduction_factor for _ in dw_kernel_sizes]
        else:
            self.dw_channels = [c // channels_reduction_factor for c in dw_channels]
        self.res_kernel_sizes = res_kernel_sizes
        self.dw_kernel_sizes = dw_kernel_sizes
        self.padding_type = padding_type
        self.use_bias = use_bias
        self.norm_layer = norm_layer
        self.norm_kwargs = norm_kwargs
        self.dropout_rate = dropout_rate
        self.active_fn = active_fn

        if self.padding_type == 'reflect':
            self.pad = cnn.ReflectionPad2d
        elif self.padding_type == 'replicate':
            self.pad = cnn.ReplicationPad2d
        elif self.padding_type == 'zero':
            self.pad = functools.partial(cnn.ZeroPad2d, value=0.0)
        else:
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)

        self.res_ops, self.dw_ops, self.pw_bn = self._build()

    def _build(self):
        _norm_kwargs = self.norm_kwargs if self.norm_kwargs is not None else {}
        res_ops = nn.ModuleList()
        for midp, k in zip(self.res_channels, self.res_kernel_sizes):
            if midp == 0:
                continue
            res_ops.append(cnn.FusedSequential(
                self.pad((k - 1) // 2),
                ConvBNReLU(self.input_dim, midp, kernel_size=k, use_bias=self.use_bias, norm_layer=self.norm_layer,
                           norm_kwargs=_norm_kwargs, active_fn=self.active_fn),
                cnn.Dropout(self.dropout_rate),
                self.pad((k - 1) // 2),
                cnn.Conv2d(midp, self.input_dim, k, 1, 0, bias=self.use_bias)))
        dw_ops = nn.ModuleList()
        for midp, k in zip(self.dw_channels, self.dw_kernel_sizes):
            if midp == 0:
                continue
            dw_ops.append(cnn.FusedSequential(
                ConvBNReLU(self.input_dim, midp, kernel_size=1, use_bias=self.use_bias, norm_layer=self.norm_layer,
                           norm_kwargs=_norm_kwargs, active_fn=self.active_fn),
                self.pad((k - 1) // 2),
                ConvBNReLU(midp, midp, kernel_size=k, groups=midp, use_bias=self.use_bias, norm_layer=self.norm_layer,
                           norm_kwargs=_norm_kwargs, active_fn=self.active_fn),
                cnn.Dropout(self.dropout_rate),
                cnn.Conv2d(midp, self.input_dim, 1, 1, 0, bias=self.use_bias)))
        pw_bn = self.norm_layer(self.input_dim, **_norm_kwargs)
        return res_ops, dw_ops, pw_bn

    # -- accessors used by prune / shrink (reference inception_modules.py:182-228) --------------------
    def get_first_res_bn(self):
        return list(self.get_named_first_res_bn().values())

    def get_first_dw_bn(self):
        return list(self.get_named_first_dw_bn().values())

    def get_first_bn(self):
        return self.get_first_res_bn() + self.get_first_dw_bn()

    def get_named_first_res_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.res_ops):
            assert isinstance(op[1], ConvBNReLU)
            res[add_prefix(f'res_ops.{i}.1.1', prefix)] = op[1][1]
        return res

    def get_named_first_dw_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.dw_ops):
            assert isinstance(op[0], ConvBNReLU)
            res[add_prefix(f'dw_ops.{i}.0.1', prefix)] = op[0][1]
        return res

    def get_named_first_bn(self, prefix=None):
        return collections.OrderedDict(list(self.get_named_first_res_bn().items()) + list(self.get_named_first_dw_bn().items()))

    def forward(self, x):
        nb = len(self.res_ops) + len(self.dw_ops)
        if nb == 0:
            return x
        if frozen.applicable(self, x):      # eval + no_grad + BatchNorm: the frozen teacher's algebraically fused block
            return frozen.block_forward(self, x)
        if fused_block.applicable(self, x):     # train-mode norms: 9 launches per block, norms folded into producers / consumers
            return fused_block.apply(self, x)
        # one alias of x per consumer (branches + residual); their gradients are summed by one add_n kernel
        xs = ops.fanout(x, nb + 1)
        branch_ops = list(self.res_ops) + list(self.dw_ops)
        if ops.branch_streams_enabled() and x.is_cuda and nb > 1:
            branches = ops.run_on_side_streams(branch_ops, xs[:nb])
        else:
            branches = [op(xi) for op, xi in zip(branch_ops, xs[:nb])]
        tmp = branches[0] if nb == 1 else ops.AddNFn.apply(*branches)
        tmp = self.pw_bn(tmp)
        return ops.AddNFn.apply(xs[nb], tmp)

    def __repr__(self):
        return ('{}({}, {}, res_channels={}, dw_channels={}, res_kernel_sizes={}, dw_kernel_sizes={})').format(
            self._get_name(), self.input_dim, self.input_dim, self.res_channels, self.dw_channels, self.res_kernel_sizes,
            self.dw_kernel_sizes)


def _get_named_block_list(m, spade=False, num_upsampling_layers=None):
    """Get `{name: module}` dictionary for inverted residual blocks (reference inception_modules.py:260-277)."""
    if spade:
        blocks = [('head_0', m.head_0)]
        blocks += [(f'G_middle_{i}', getattr(m, f'G_middle_{i}')) for i in range(2)]
        blocks += [(f'up_{i}', getattr(m, f'up_{i}')) for i in range(4)]
        if num_upsampling_layers == 'most':
            blocks += [('up_4', m.up_4)]
        return collections.OrderedDict(blocks)
    return collections.OrderedDict(('features.{}'.format(name), block) for name, block in m.features.named_children())


def output_network(model):
    """Output network kwargs in `searched_network` style (reference inception_modules.py:246-258)."""
    res = []
    for block in model.get_named_block_list().values():
        res.append([block.input_dim, block.res_channels, block.dw_channels, block.res_kernel_sizes, block.dw_kernel_sizes,
                    getattr(block, 'stride', 1)])
    return {'inverted_residual_setting': res}


# ------------------------------------------------------------------------------------------------ GauGAN / SPADE blocks
def _branch_channels(channels, default, factor, kernel_sizes):
    if channels is None:
        return [default // factor for _ in kernel_sizes]
    if type(channels) == int:
        return [channels // factor for _ in kernel_sizes]
    assert len(channels) == len(kernel_sizes)
    return [c // factor for c in channels]


class ConvSyncBNReLU(nn.Module):
    """Conv2d(pad same) -> norm -> activation with sub-modules `conv`, `norm`, `active` (reference inception_modules.py:280-313).
    Train mode: conv kernel + split-phase batch norm with the activation fused into its apply pass; frozen (eval, no grad):
    the running statistics are folded into the conv and the whole thing is ONE kernel."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1, use_bias=True, norm_layer=None, active_fn=None,
                 spectral_norm=False, spade=False):
        super().__init__()
        self.conv = cnn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=use_bias)
        self.norm = norm_layer(norm_nc=out_planes) if spade else norm_layer(out_planes)
        self.active = active_fn()
        if spectral_norm:
            self.conv = cnn.spectral_norm(self.conv)

    def forward(self, x, seg=None):
        if seg is not None:
            return self.active(self.norm(self.conv(x), seg))
        if cnn._bn_folds(self.norm) and 'weight_orig' not in self.conv._parameters:
            y = self.conv(x, fuse_act=self.active, fold_bn=self.norm)
            return self.active(self.norm(y, applied=True), applied=True)
        y = self.norm(self.conv(x), fuse_act=self.active)
        return self.active(y, applied=True)

    def remove_spectral_norm(self):
        self.conv = cnn.remove_spectral_norm(self.conv)


class Conv(nn.Module):
    """Convolution with "same" padding (reference inception_modules.py:316-341)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1, use_bias=True, spectral_norm=False):
        super().__init__()
        self.conv = cnn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=use_bias)
        if spectral_norm:
            self.conv = cnn.spectral_norm(self.conv)

    def forward(self, x):
        return self.conv(x)

    def remove_spectral_norm(self):
        self.conv = cnn.remove_spectral_norm(self.conv)


def _run_branches(branch_ops, x, extra=()):
    """sum of the (independent) branches applied to x, plus `extra` addends, in ONE add_n kernel; the branches run on side HIP
    streams (ops.run_on_side_streams) so their small kernels overlap."""
    nb = len(branch_ops)
    xs = ops.fanout(x, nb) if x.requires_grad and torch.is_grad_enabled() else [x] * nb
    if ops.branch_streams_enabled() and x.is_cuda and nb > 1:
        outs = ops.run_on_side_streams(branch_ops, xs)
    else:
        outs = [op(xi) for op, xi in zip(branch_ops, xs)]
    outs = list(outs) + list(extra)
    return outs[0] if len(outs) == 1 else ops.AddNFn.apply(*outs)


def seg_at(segmap, size):
    """F.interpolate(segmap, size, mode='nearest') (inception_modules.py:748), computed once per (segmap, resolution): every SPADE
    layer of the teacher AND the student that runs at this resolution shares the result."""
    size = (int(size[0]), int(size[1]))
    if tuple(segmap.shape[2:]) == size:
        return segmap
    cache = getattr(segmap, '_cat_pyramid', None)
    if cache is None or cache[0] != segmap._version:
        cache = (segmap._version, {})
        segmap._cat_pyramid = cache
    out = cache[1].get(size)
    if out is None:
        with torch.no_grad():
            out = ops.interp_nearest(segmap, size)
        cache[1][size] = out
    return out


class InceptionSPADE(nn.Module):
    """SPADE with inception-style gamma/beta branches (reference inception_modules.py:564-769):
    out = param_free_norm(x) * (1 + gamma) + beta, [gamma | beta] = sum_k res_k(seg) + sum_k dw_k(seg)."""

    def __init__(self, norm, norm_nc, label_nc, nhidden=128, opt=None):
        super(InceptionSPADE, self).__init__()
        res_kernel_sizes = opt.kernel_sizes
        dw_kernel_sizes = opt.kernel_sizes
        if type(res_kernel_sizes) == int:
            res_kernel_sizes = dw_kernel_sizes = [res_kernel_sizes]
        self.norm_layer = functools.partial(cnn.SynchronizedBatchNorm2d, affine=True)
        self.active_fn = functools.partial(cnn.ReLU, inplace=True)
        self.param_free_norm_layer = norm
        self.input_dim = label_nc
        self.output_dim = norm_nc
        self.res_channels = _branch_channels(opt.channels, nhidden, opt.channels_reduction_factor, res_kernel_sizes)
        self.dw_channels = _branch_channels(opt.channels, nhidden, opt.channels_reduction_factor, dw_kernel_sizes)
        self.res_kernel_sizes = res_kernel_sizes
        self.dw_kernel_sizes = dw_kernel_sizes
        self.param_free_norm, self.res_ops, self.dw_ops = self._build()

    def _build(self):
        param_free_norm = self.param_free_norm_layer(self.output_dim, affine=False)
        res_ops = nn.ModuleList()
        for midp, k in zip(self.res_channels, self.res_kernel_sizes):
            if midp == 0:
                continue
            res_ops.append(nn.Sequential(
                ConvSyncBNReLU(self.input_dim, midp, kernel_size=k, norm_layer=self.norm_layer, active_fn=self.active_fn),
                cnn.Conv2d(midp, 2 * self.output_dim, kernel_size=k, padding=(k - 1) // 2)))
        dw_ops = nn.ModuleList()
        for midp, k in zip(self.dw_channels, self.dw_kernel_sizes):
            if midp == 0:
                continue
            dw_ops.append(nn.Sequential(
                ConvSyncBNReLU(self.input_dim, midp, kernel_size=1, norm_layer=self.norm_layer, active_fn=self.active_fn),
                ConvSyncBNReLU(midp, midp, kernel_size=k, groups=midp, norm_layer=self.norm_layer, active_fn=self.active_fn),
                cnn.Conv2d(midp, 2 * self.output_dim, kernel_size=1)))
        return param_free_norm, res_ops, dw_ops

    def get_first_res_bn(self):
        return list(self.get_named_first_res_bn().values())

    def get_first_dw_bn(self):
        return list(self.get_named_first_dw_bn().values())

    def get_first_bn(self):
        return self.get_first_res_bn() + self.get_first_dw_bn()

    def get_named_first_res_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.res_ops):
            assert isinstance(op[0], ConvSyncBNReLU)
            res[add_prefix(f'res_ops.{i}.0.norm', prefix)] = op[0].norm
        return res

    def get_named_first_dw_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.dw_ops):
            assert isinstance(op[0], ConvSyncBNReLU)
            res[add_prefix(f'dw_ops.{i}.0.norm', prefix)] = op[0].norm
        return res

    def get_named_first_bn(self, prefix=None):
        return collections.OrderedDict(list(self.get_named_first_res_bn().items()) + list(self.get_named_first_dw_bn().items()))

    def forward(self, x, segmap, fuse_act=None):
        """`fuse_act`: the activation SPADEInvertedResidualChannels applies right after (inception_modules.py:553), fused here."""
        pfn = self.param_free_norm
        instance = isinstance(pfn, cnn.InstanceNorm2d)
        if not instance and not isinstance(pfn, cnn.BatchNorm2d):
            raise NotImplementedError('SPADE param-free norm: (sync)batch or instance')
        if instance and pfn.track_running_stats:
            raise NotImplementedError('InstanceNorm2d(track_running_stats=True) is not what norm_G = spadeinstance builds')
        act, slope = cnn._act_code(fuse_act)
        seg = seg_at(segmap, x.shape[2:])
        branch_ops = list(self.res_ops) + list(self.dw_ops)
        if not branch_ops:      # gamma = beta = 0: the plain param-free norm
            return pfn(x, fuse_act=fuse_act)
        from . import fused_spade
        pre = self.__dict__.pop('_cat_gb_pre', None)      # computed by the generator's pre-pass (all SPADE layers in lockstep, N > 1 ranks)
        if pre is not None:      # (gamma|beta, data pointer, version) of the map it was computed from: a stale entry never meets another input
            gb_pre, src_ptr, src_ver = pre
            pre = gb_pre if (src_ptr == segmap.data_ptr() and src_ver == segmap._version and
                             tuple(gb_pre.shape[2:]) == tuple(seg.shape[2:])) else None
        if pre is not None:
            gb = pre
        elif fused_spade._UNITS in ('all', 'gb') and fused_spade.applicable(self.res_ops, self.dw_ops, seg, self.training):
            # all first convs / norms / depthwise convs / the 2C-channel branch sum of the gamma|beta net as 5 launches (cat_amd/fused_spade.py)
            gb = fused_spade.apply(self, '_cat_fused_gb', self.res_ops, self.dw_ops, self.input_dim, 2 * self.output_dim, seg)
        else:
            gb = _run_branches(branch_ops, seg)
        if instance:      # norm_G = 'spadeinstance...' (reference :414-415): per-image statistics in train and eval mode alike
            return ops.SpadeInstanceFn.apply(x, gb, float(pfn.eps), act, slope)
        if pfn.training or not pfn.track_running_stats:
            track = pfn.training and pfn.track_running_stats
            return ops.SpadeFn.apply(x, gb, pfn.running_mean if track else None, pfn.running_var if track else None, float(pfn.eps),
                                     float(pfn.momentum), act, slope)
        return ops.spade_eval(x, gb, pfn.running_mean, pfn.running_var, float(pfn.eps), act, slope)

    def __repr__(self):
        return ('{}({}, {}, res_channels={}, dw_channels={}, res_kernel_sizes={}, dw_kernel_sizes={})').format(
            self._get_name(), self.input_dim, self.input_dim, self.res_channels, self.dw_channels, self.res_kernel_sizes,
            self.dw_kernel_sizes)


class SPADEInvertedResidualChannels(nn.Module):
    """act(SPADE(x, seg)) -> sum_k res_k + sum_k dw_k, plus (learned) shortcut (reference inception_modules.py:344-562)."""

    def __init__(self, fin, fout, opt):
        super().__init__()
        self.opt = opt
        self.learned_shortcut = (fin != fout)
        fmiddle = min(fin, fout)
        res_kernel_sizes = opt.kernel_sizes
        dw_kernel_sizes = opt.kernel_sizes
        if type(res_kernel_sizes) == int:
            res_kernel_sizes = dw_kernel_sizes = [res_kernel_sizes]
        self.input_dim = fin
        self.output_dim = fout
        self.res_channels = _branch_channels(opt.channels, fmiddle, opt.channels_reduction_factor, res_kernel_sizes)
        self.dw_channels = _branch_channels(opt.channels, fmiddle, opt.channels_reduction_factor, dw_kernel_sizes)
        self.res_kernel_sizes = res_kernel_sizes
        self.dw_kernel_sizes = dw_kernel_sizes
        self.active_fn = get_active_fn(opt.active_fn)
        self.active = self.active_fn()
        self.spectral_norm = 'spectral' in opt.norm_G
        spade_config_str = opt.norm_G.replace('spectral', '')
        if not spade_config_str.startswith('spade'):
            raise NotImplementedError
        parsed = re.search(r'spade(\D+)(\d)x\d', spade_config_str)
        param_free_norm_type = str(parsed.group(1))
        self.spade_norm = InceptionSPADE
        if param_free_norm_type == 'syncbatch':
            self.norm_layer = cnn.SynchronizedBatchNorm2d
        elif param_free_norm_type == 'batch':
            self.norm_layer = cnn.BatchNorm2d
        elif param_free_norm_type == 'instance':
            self.norm_layer = cnn.InstanceNorm2d
        else:
            raise ValueError(f'{param_free_norm_type} is not a recognized param-free norm type in SPADE')
        self.semantic_nc = opt.semantic_nc
        self.res_ops, self.dw_ops, self.shortcut, self.spade = self._build()

    def _build(self, build_only=False):
        sn = self.spectral_norm
        res_ops = nn.ModuleList()
        for midp, k in zip(self.res_channels, self.res_kernel_sizes):
            if midp == 0:
                continue
            res_ops.append(nn.Sequential(
                ConvSyncBNReLU(self.input_dim, midp, kernel_size=k, norm_layer=functools.partial(self.norm_layer, affine=True),
                               active_fn=self.active_fn, spectral_norm=sn),
                Conv(midp, self.output_dim, kernel_size=k, spectral_norm=sn)))
        dw_ops = nn.ModuleList()
        for midp, k in zip(self.dw_channels, self.dw_kernel_sizes):
            if midp == 0:
                continue
            dw_ops.append(nn.Sequential(
                ConvSyncBNReLU(self.input_dim, midp, kernel_size=1, norm_layer=functools.partial(self.norm_layer, affine=True),
                               active_fn=self.active_fn, spectral_norm=sn),
                ConvSyncBNReLU(midp, midp, kernel_size=k, groups=midp, norm_layer=functools.partial(self.norm_layer, affine=False),
                               active_fn=self.active_fn, spectral_norm=sn),
                Conv(midp, self.output_dim, kernel_size=1, spectral_norm=sn)))
        if self.learned_shortcut:
            shortcut = nn.Sequential(self.norm_layer(self.input_dim, affine=True),
                                     Conv(self.input_dim, self.output_dim, kernel_size=1, use_bias=False, spectral_norm=sn))
        else:
            shortcut = None
        if build_only:
            self.spade.param_free_norm, self.spade.res_ops, self.spade.dw_ops = self.spade._build()
            spade = self.spade
        else:
            spade = self.spade_norm(norm=self.norm_layer, norm_nc=self.input_dim, label_nc=self.semantic_nc, opt=self.opt)
        return res_ops, dw_ops, shortcut, spade

    def get_first_res_bn(self):
        return list(self.get_named_first_res_bn().values())

    def get_first_dw_bn(self):
        return list(self.get_named_first_dw_bn().values())

    def get_first_bn(self):
        return self.get_first_res_bn() + self.get_first_dw_bn()

    def get_named_first_res_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.res_ops):
            assert isinstance(op[0], ConvSyncBNReLU)
            assert isinstance(op[0].norm, self.norm_layer)
            res[add_prefix(f'res_ops.{i}.0.norm', prefix)] = op[0].norm
        return res

    def get_named_first_dw_bn(self, prefix=None):
        res = collections.OrderedDict()
        for i, op in enumerate(self.dw_ops):
            assert isinstance(op[0], ConvSyncBNReLU)
            assert isinstance(op[0].norm, self.norm_layer)
            res[add_prefix(f'dw_ops.{i}.0.norm', prefix)] = op[0].norm
        return res

    def get_named_first_bn(self, prefix=None):
        return collections.OrderedDict(list(self.get_named_first_res_bn().items()) + list(self.get_named_first_dw_bn().items()))

    def _shortcut(self, x):
        if self.shortcut is None:
            return x
        return self.shortcut[1](self.shortcut[0](x))

    def forward(self, x, seg):
        branch_ops = list(self.res_ops) + list(self.dw_ops)
        if not branch_ops:
            return self._shortcut(x)
        grad = x.requires_grad and torch.is_grad_enabled()
        x_spade, x_short = ops.fanout(x, 2) if grad else (x, x)
        tmp = self.spade(x_spade, seg, fuse_act=self.active)
        tmp = self.active(tmp, applied=True)
        from . import fused_spade
        if fused_spade._UNITS in ('all', 'main') and fused_spade.applicable(self.res_ops, self.dw_ops, tmp, self.training):
            return fused_spade.apply(self, '_cat_fused_main', self.res_ops, self.dw_ops, self.input_dim, self.output_dim, tmp,
                                     addend=self._shortcut(x_short))
        return _run_branches(branch_ops, tmp, extra=(self._shortcut(x_short),))

    def remove_spectral_norm(self):
        """Reference inception_modules.py:571-586 (export path): every conv of every branch and of the learned shortcut gets its
        normalised weight baked in; a conv that carries no spectral norm raises ValueError, as torch's remove_spectral_norm does."""
        carriers = [m for op in list(self.res_ops) + list(self.dw_ops) for m in op]
        if self.shortcut is not None:
            carriers.append(self.shortcut[1])
        for m in carriers:
            if not isinstance(m, (ConvSyncBNReLU, Conv)):
                raise TypeError('unexpected module in a SPADE branch: %s' % type(m).__name__)
            m.remove_spectral_norm()

    def __repr__(self):
        return ('{}({}, {}, res_channels={}, dw_channels={}, res_kernel_sizes={}, dw_kernel_sizes={})\n\tSPADE: {}').format(
            self._get_name(), self.input_dim, self.input_dim, self.res_channels, self.dw_channels, self.res_kernel_sizes,
            self.dw_kernel_sizes, self.spade)
