"""Exporter-ready twin of a cat_amd generator.

The reference exports the trained student with `torch.onnx.export(model.netG_student.cpu(), rand_input.cpu(), ...)`
(onnx_exporter.py:142-163).  cat_amd's generators are HIP-only: NHWC activations, conv weights in the kernels' padded channels-last
storage, no CPU arithmetic.  `to_reference_module(net)` builds a STOCK torch.nn module tree on the host -- NCHW, dense OIHW weights,
plain nn.Conv2d / nn.BatchNorm2d / nn.InstanceNorm2d / nn.ReflectionPad2d ... -- with

  * the reference's module names, hence the reference's `state_dict` keys and shapes (a checkpoint saved from the twin loads into
    the reference's `define_G` network and back);
  * forwards written in plain torch ops that `torch.onnx.export` / `torch.jit.trace` can follow on CPU.

The twin is a conversion target, not an execution path: nothing in the training / evaluation step uses it.
Topologies restated: InceptionGenerator.forward (models/modules/inception_architecture/inception_generator.py:137-142),
InvertedResidualChannels.forward (models/modules/inception_modules.py:230-236), InceptionSPADEGenerator.forward
(inception_spade_generator.py:63-124), SPADEInvertedResidualChannels.forward (inception_modules.py:549-562), InceptionSPADE.forward
(:746-762), ConvSyncBNReLU / Conv (:280-341)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn as cnn
from .inception_generator import InceptionGenerator
from .inception_modules import Conv, ConvSyncBNReLU, InceptionSPADE, InvertedResidualChannels, SPADEInvertedResidualChannels
from .inception_spade_generator import InceptionSPADEGenerator


def _dense(t):
    """A parameter / buffer as a dense host tensor in its logical (reference) layout."""
    return t.detach().to('cpu').contiguous().clone()


# ---------------------------------------------------------------------------------------------------------------- composite twins
class TwinInvertedResidualChannels(nn.Module):
    """x + pw_bn(sum_k res_k(x) + sum_k dw_k(x)); attribute names = the reference's (res_ops, dw_ops, pw_bn)."""

    def __init__(self, res_ops, dw_ops, pw_bn):
        super().__init__()
        self.res_ops, self.dw_ops, self.pw_bn = res_ops, dw_ops, pw_bn

    def forward(self, x):
        branches = [op(x) for op in self.res_ops] + [op(x) for op in self.dw_ops]
        if not branches:
            return x
        total = branches[0]
        for b in branches[1:]:
            total = total + b
        return x + self.pw_bn(total)


class TwinInceptionGenerator(nn.Module):
    def __init__(self, down_sampling, features, up_sampling):
        super().__init__()
        self.down_sampling, self.features, self.up_sampling = down_sampling, features, up_sampling

    def forward(self, input):
        return self.up_sampling(self.features(self.down_sampling(input)))


class TwinConvNormAct(nn.Module):
    """conv ("same" zero padding) -> norm -> activation with the reference's sub-module names conv / norm / active."""

    def __init__(self, conv, norm, active):
        super().__init__()
        self.conv, self.norm, self.active = conv, norm, active

    def forward(self, x):
        return self.active(self.norm(self.conv(x)))


class TwinConv(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.conv = conv

    def forward(self, x):
        return self.conv(x)


def _branch_sum(res_ops, dw_ops, x):
    outs = [op(x) for op in res_ops] + [op(x) for op in dw_ops]
    total = outs[0]
    for o in outs[1:]:
        total = total + o
    return total


class TwinInceptionSPADE(nn.Module):
    """param_free_norm(x) * (1 + gamma) + beta with [gamma | beta] = the six-branch net applied to the resized segmentation map."""

    def __init__(self, param_free_norm, res_ops, dw_ops):
        super().__init__()
        self.param_free_norm, self.res_ops, self.dw_ops = param_free_norm, res_ops, dw_ops

    def forward(self, x, segmap):
        normalized = self.param_free_norm(x)
        if len(self.res_ops) + len(self.dw_ops) == 0:
            return normalized
        seg = F.interpolate(segmap, size=x.shape[2:], mode='nearest')
        gb = _branch_sum(self.res_ops, self.dw_ops, seg)
        c = x.shape[1]
        return normalized * (1 + gb[:, :c]) + gb[:, c:]


class TwinSPADEInvertedResidualChannels(nn.Module):
    def __init__(self, spade, active, res_ops, dw_ops, shortcut):
        super().__init__()
        self.active = active
        self.res_ops, self.dw_ops = res_ops, dw_ops      # registration order = the reference's (state_dict key order)
        if shortcut is not None:
            self.shortcut = shortcut
        self.spade = spade

    def forward(self, x, seg):
        short = self.shortcut(x) if hasattr(self, 'shortcut') else x
        if len(self.res_ops) + len(self.dw_ops) == 0:
            return short
        t = self.active(self.spade(x, seg))
        return _branch_sum(self.res_ops, self.dw_ops, t) + short


class TwinInceptionSPADEGenerator(nn.Module):
    def __init__(self, src, convert):
        super().__init__()
        self.sw, self.sh = src.sw, src.sh
        self.num_upsampling_layers = src.opt.num_upsampling_layers
        self.fc_norm = convert(src.fc_norm)      # registration order = the reference's (inception_spade_generator.py:14-45): state_dict key order
        self.fc = convert(src.fc)
        names = ['head_0', 'G_middle_0', 'G_middle_1', 'up_0', 'up_1', 'up_2', 'up_3'] + (['up_4'] if self.num_upsampling_layers == 'most' else [])
        for name in names:
            setattr(self, name, convert(getattr(src, name)))
        self.conv_img = convert(src.conv_img)
        self.up = nn.Upsample(scale_factor=2)

    def forward(self, input):
        seg = input
        x = F.interpolate(seg, size=(self.sh, self.sw))
        x = self.fc_norm(self.fc(x))
        x = self.head_0(x, seg)
        x = self.up(x)
        x = self.G_middle_0(x, seg)
        if self.num_upsampling_layers in ('more', 'most'):
            x = self.up(x)
        x = self.G_middle_1(x, seg)
        for name in ('up_0', 'up_1', 'up_2', 'up_3'):
            x = self.up(x)
            x = getattr(self, name)(x, seg)
        if self.num_upsampling_layers == 'most':
            x = self.up(x)
            x = self.up_4(x, seg)
        return torch.tanh(self.conv_img(F.leaky_relu(x, 2e-1)))


# ---------------------------------------------------------------------------------------------------------------- leaves
def _conv(m, cls):
    if 'weight_orig' in m._parameters:
        raise NotImplementedError('to_reference_module: the conv still carries spectral norm -- call remove_spectral_norm() first, '
                                  'as the export path does (inception_modules.py:571-586)')
    kw = dict(kernel_size=m.kernel_size, stride=m.stride, padding=m.padding, dilation=m.dilation, groups=m.groups, bias=m.bias is not None)
    if cls is nn.ConvTranspose2d:
        kw['output_padding'] = m.output_padding
    else:
        kw['padding_mode'] = m.padding_mode
    t = cls(m.in_channels, m.out_channels, **kw)
    with torch.no_grad():
        t.weight.copy_(_dense(m.weight))
        if m.bias is not None:
            t.bias.copy_(_dense(m.bias))
    return t


def _norm(m):
    if isinstance(m, cnn.InstanceNorm2d):
        t = nn.InstanceNorm2d(m.num_features, eps=m.eps, momentum=m.momentum, affine=m.affine, track_running_stats=m.track_running_stats)
    else:
        # SynchronizedBatchNorm2d on one device / in eval mode IS F.batch_norm (sync_batchnorm/batchnorm.py:69-72), and it derives from torch's
        # _BatchNorm: same state_dict keys as nn.BatchNorm2d
        t = nn.BatchNorm2d(m.num_features, eps=m.eps, momentum=m.momentum, affine=m.affine, track_running_stats=m.track_running_stats)
    with torch.no_grad():
        for name in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
            src = getattr(m, name, None)
            dst = getattr(t, name, None)
            if src is not None and dst is not None:
                dst.copy_(_dense(src))
    return t


def _leaf(m):
    if isinstance(m, cnn.ConvTranspose2d):
        return _conv(m, nn.ConvTranspose2d)
    if isinstance(m, cnn.Conv2d):
        return _conv(m, nn.Conv2d)
    if isinstance(m, (cnn.BatchNorm2d, cnn.InstanceNorm2d)):
        return _norm(m)
    if isinstance(m, cnn.ReflectionPad2d):
        return nn.ReflectionPad2d(m.padding)
    if isinstance(m, cnn.ReplicationPad2d):
        return nn.ReplicationPad2d(m.padding)
    if isinstance(m, cnn.ZeroPad2d):
        return nn.ZeroPad2d(m.padding)
    if isinstance(m, cnn.LeakyReLU):
        return nn.LeakyReLU(m.negative_slope)
    if isinstance(m, cnn.ReLU6):
        return nn.ReLU6()
    if isinstance(m, cnn.ReLU):
        return nn.ReLU()
    if isinstance(m, cnn.Tanh):
        return nn.Tanh()
    if isinstance(m, cnn.Dropout):
        return nn.Dropout(m.p)
    if isinstance(m, (cnn.Identity, nn.Identity)):
        return nn.Identity()
    if isinstance(m, cnn.Upsample):
        return nn.Upsample(scale_factor=m.scale_factor, mode=m.mode)
    return None


def _has_own_weight_state(m):
    return any(True for _ in m.parameters(recurse=False)) or any(True for _ in m.buffers(recurse=False))


def _convert(m):
    leaf = _leaf(m)
    if leaf is not None:
        return leaf
    if isinstance(m, InceptionGenerator):
        return TwinInceptionGenerator(_convert(m.down_sampling), _convert(m.features), _convert(m.up_sampling))
    if isinstance(m, InvertedResidualChannels):
        return TwinInvertedResidualChannels(_convert(m.res_ops), _convert(m.dw_ops), _convert(m.pw_bn))
    if isinstance(m, InceptionSPADEGenerator):
        return TwinInceptionSPADEGenerator(m, _convert)
    if isinstance(m, SPADEInvertedResidualChannels):
        return TwinSPADEInvertedResidualChannels(_convert(m.spade), _convert(m.active), _convert(m.res_ops), _convert(m.dw_ops),
                                                 None if m.shortcut is None else _convert(m.shortcut))
    if isinstance(m, InceptionSPADE):
        return TwinInceptionSPADE(_convert(m.param_free_norm), _convert(m.res_ops), _convert(m.dw_ops))
    if isinstance(m, ConvSyncBNReLU):
        return TwinConvNormAct(_convert(m.conv), _convert(m.norm), _convert(m.active))
    if isinstance(m, Conv):
        return TwinConv(_convert(m.conv))
    if isinstance(m, nn.ModuleList):
        return nn.ModuleList([_convert(c) for c in m])
    if isinstance(m, nn.Sequential):      # FusedSequential, ConvBNReLU: same child names (indices)
        if _has_own_weight_state(m):
            raise NotImplementedError('to_reference_module: a container with parameters of its own: %s' % type(m).__name__)
        return nn.Sequential(*[_convert(c) for c in m])
    raise NotImplementedError('to_reference_module: no stock-torch twin for %s' % type(m).__name__)


def to_reference_module(net):
    """Stock torch.nn twin of a cat_amd generator (InceptionGenerator / InceptionSPADEGenerator or any sub-tree of one): host tensors, NCHW,
    the reference's state_dict keys and shapes, train / eval flags copied module by module.  Values are copied, not shared."""
    twin = _convert(net)
    src = dict(net.named_modules())
    for name, t in twin.named_modules():
        s = src.get(name)
        t.train(s.training if s is not None else net.training)
    a, b = list(net.state_dict().keys()), list(twin.state_dict().keys())
    if a != b:
        raise RuntimeError('to_reference_module: state_dict keys differ: %s' % sorted(set(a) ^ set(b))[:8])
    return twin


def export_onnx(net, example_input, path, **kw):
    """What onnx_exporter.py:142-163 does with the student, through the twin (needs the `onnx` package, as the reference does)."""
    twin = to_reference_module(net).eval()
    args = dict(export_params=True, opset_version=11, do_constant_folding=True, input_names=['input'], output_names=['output'],
                dynamic_axes={'input': {0: 'batch_size'}, 'output': {0: 'batch_size'}})
    args.update(kw)
    torch.onnx.export(twin, example_input.detach().cpu(), path, **args)
    return twin
