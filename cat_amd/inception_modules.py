"""Student / teacher building blocks: same classes, constructor arguments, attribute names and state_dict keys as
the reference's models/modules/inception_modules.py:12-243 (get_active_fn, ConvBNReLU, InvertedResidualChannels),
built from cat_amd.nn layers so every op is a gfx950 kernel."""
import collections
import functools

from torch import nn

from . import nn as cnn
from . import ops


def add_prefix(name, prefix=None, split='.'):
    """reference common.py add_prefix."""
    if prefix is not None:
        return '{}{}{}'.format(prefix, split, name)
    return name


def get_active_fn(name):
    """reference inception_modules.py:12-19 (ReLU6 is never selected by the distillation scripts)."""
    active_fn = {
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


def _get_named_block_list(m):
    """Get `{name: module}` dictionary for inverted residual blocks (reference inception_modules.py `_get_named_block_list`)."""
    blocks = list(m.features.named_children())
    all_blocks = []
    for name, block in blocks:
        all_blocks.append(('features.{}'.format(name), block))
    return collections.OrderedDict(all_blocks)
