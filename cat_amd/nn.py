"""Layer classes with the reference's names, constructor signatures, state_dict keys and default initialisation
(they subclass the torch.nn classes CAT instantiates) whose forward runs the gfx950 kernels of libcat_hip.so.

Fusion happens at call time, not by rewriting the module tree, so `shrink_model`-style surgery on
`down_sampling[idx]` and forward hooks on any sub-module keep working exactly as in the reference:
  ReflectionPad2d  -> returns a lazy `Padded` marker that the next conv folds into its im2col gather;
  Conv / Norm      -> accept `fuse_act=` from `FusedSequential`, the activation module that follows is then called
                      with `applied=True` (identity) so its hooks (e.g. 'down_sampling.9') still fire.
"""
import os

import torch
from torch import nn

from . import _lib as L
from . import ops
from . import optim


class Padded:
    """Activation + pending reflection padding (consumed by the next convolution)."""
    __slots__ = ('x', 'pad', 'mode')

    def __init__(self, x, pad, mode):
        self.x, self.pad, self.mode = x, pad, mode


def _act_code(m):
    if m is None:
        return L.ACT_NONE, 0.0
    if isinstance(m, nn.LeakyReLU):
        return L.ACT_LRELU, float(m.negative_slope)
    if isinstance(m, nn.ReLU6):
        return L.ACT_RELU6, 0.0
    if isinstance(m, nn.ReLU):
        return L.ACT_RELU, 0.0
    if isinstance(m, nn.Tanh):
        return L.ACT_TANH, 0.0
    raise NotImplementedError('activation %s has no fused kernel' % type(m).__name__)


class _Act:
    def forward(self, x, applied=False):
        if applied:
            return x
        code, slope = _act_code(self)
        return ops.ActFn.apply(x, code, slope)


class ReLU(_Act, nn.ReLU):
    pass


class LeakyReLU(_Act, nn.LeakyReLU):
    pass


class Tanh(_Act, nn.Tanh):
    pass


class ReLU6(_Act, nn.ReLU6):
    pass


_ACTS = (ReLU, LeakyReLU, Tanh, ReLU6)


class Identity(nn.Module):
    def forward(self, x):
        return x


class Dropout(nn.Dropout):
    """The distillation scripts all run with dropout_rate 0; a non-zero rate is not on the path."""

    def forward(self, x):
        if self.p != 0 and self.training:
            raise NotImplementedError('dropout with p > 0 is outside the accelerated hot path')
        return x


class ReflectionPad2d(nn.ReflectionPad2d):
    def forward(self, x):
        p = self.padding[0]
        if any(q != p for q in self.padding):
            raise NotImplementedError('only symmetric reflection padding is supported')
        if isinstance(x, Padded):
            raise NotImplementedError('stacked paddings')
        return Padded(x, p, L.PAD_REFLECT) if p > 0 else x      # x may carry a pending norm (ops.Normed): FusedSequential resolves it


class ReplicationPad2d(nn.ReplicationPad2d):
    """padding_type='replicate' (reference inception_modules.py:114-115): unlike the reflect / zero paddings this one is not folded into
    the next convolution's gather -- the padded copy is written (the option is not used by any launch script)."""

    def forward(self, x):
        p = self.padding[0]
        if any(q != p for q in self.padding):
            raise NotImplementedError('only symmetric replication padding is supported')
        if isinstance(x, Padded):
            raise NotImplementedError('stacked paddings')
        return ops.ReplicatePadFn.apply(x, p) if p > 0 else x


class ZeroPad2d(nn.ConstantPad2d):
    def __init__(self, padding, value=0.0):
        super().__init__(padding, value)

    def forward(self, x):
        p = self.padding[0]
        if any(q != p for q in self.padding) or self.value != 0:
            raise NotImplementedError('only symmetric zero padding is supported')
        return Padded(x, p, L.PAD_ZERO) if p > 0 else x


def _to_channels_last_(module):
    """Re-house a conv weight as [O][kh][kw][round_up(I,4)] (zero padded channels_last); logical values are unchanged."""
    w = module.weight
    if w.dim() == 4 and w.shape[1] > 1 and ops.weight_wcs(w) != ops.cs_for(w.shape[1]) and w.is_cuda:
        _rehouse_(w)


def _rehouse_(w):
    """w.data -> zero-padded channels_last storage.  The old storage was allocated on the stream of model construction; this may run on ANOTHER
    stream (the frozen teacher's first forward on the data-parallel schedule's side stream): record_stream keeps the caching allocator from
    handing the old block to the construction stream's next allocation before the copy below has executed (round 6: the teacher's late layers
    were re-housed from recycled memory in ~half of the two-rank runs on one stream)."""
    old = w.data
    new = ops.padded_weight_like(w.shape, w.device)
    new.copy_(old)
    if old.is_cuda:
        old.record_stream(torch.cuda.current_stream(old.device))
    w.data = new


def _folded_eval_bn(conv, bn, out_dim):
    """Frozen-teacher fast path: conv -> BatchNorm2d(eval, running stats) is an affine map per output channel, so it is folded
    into the conv's weight / bias once (cached until any of the tensors involved changes) and the norm kernel disappears.
    One-time setup arithmetic, not part of the per-step kernel stream."""
    tensors = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    # optimizer-owned weights are updated through raw pointers (no version bump): such layers re-fold after every optimizer step
    trainable = any(getattr(t, '_cat_grad_view', None) is not None for t in tensors if t is not None)
    key = (tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors), optim.weights_epoch() if trainable else -1)
    cache = getattr(conv, '_cat_fold', None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight
        shift = -bn.running_mean * scale
        if bn.bias is not None:
            shift = shift + bn.bias
        shape = [1, 1, 1, 1]
        shape[out_dim] = -1
        w = conv.weight.detach()
        if w.dim() == 4 and w.shape[1] > 1:
            wf = ops.padded_weight_like(w.shape, w.device)
            wf.copy_(w * scale.view(shape))
        else:
            wf = (w * scale.view(shape)).contiguous()
        bf = shift if conv.bias is None else conv.bias.detach() * scale + shift
        bf = bf.contiguous()
    conv._cat_fold = (key, wf, bf)
    return wf, bf


def _bn_folds(bn):
    return isinstance(bn, BatchNorm2d) and not bn.training and bn.track_running_stats and not torch.is_grad_enabled()


class Conv2d(nn.Conv2d):
    """nn.Conv2d (dense or depthwise).  Weight storage is channels_last ([O][kh][kw][I]); state_dict values are
    unchanged (logical OIHW)."""

    def forward(self, x, fuse_act=None, fold_bn=None):
        if 'weight_orig' in self._parameters:       # spectral_norm(conv): weight = weight_orig / sigma, computed per forward
            if fold_bn is not None:
                raise NotImplementedError('spectral norm + folded BatchNorm')
            return self._conv(x, _sn_weight(self), self.bias, fuse_act)
        weight, bias = self.weight, self.bias
        if fold_bn is not None:
            weight, bias = _folded_eval_bn(self, fold_bn, 0)
        return self._conv(x, weight, bias, fuse_act)

    def _conv(self, x, weight, bias, fuse_act):
        pad, mode = self.padding[0], L.PAD_ZERO
        if isinstance(x, Padded):
            if pad != 0:
                raise NotImplementedError('explicit padding in front of a padded conv')
            x, pad, mode = x.x, x.pad, x.mode
        if self.padding[0] != self.padding[1] or self.stride[0] != self.stride[1] or self.dilation != (1, 1):
            raise NotImplementedError('conv2d: only square stride/padding and dilation 1')
        if self.padding_mode != 'zeros':
            raise NotImplementedError('conv2d: use ReflectionPad2d for reflect padding (as the reference does)')
        act, slope = _act_code(fuse_act)
        if self.groups == 1:
            if weight is self.weight:
                _to_channels_last_(self)
                weight = self.weight
            return ops.Conv2dFn.apply(x, weight, bias, self.stride[0], pad, mode, act, slope)
        if self.groups == self.in_channels == self.out_channels and self.stride[0] == 1:
            y = ops.DwConv2dFn.apply(x, weight, bias, pad, mode)
            return ops.ActFn.apply(y, act, slope) if act != L.ACT_NONE else y
        raise NotImplementedError('conv2d: groups must be 1 or == channels (depthwise, stride 1)')


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, fuse_act=None, fold_bn=None):
        if isinstance(x, Padded):
            raise NotImplementedError('padding in front of a transposed conv')
        if self.groups != 1 or self.dilation != (1, 1) or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError('conv_transpose2d: groups 1, dilation 1, square geometry only')
        if fold_bn is not None:      # frozen (eval, no grad): BatchNorm folded into the weights, activation in the kernel epilogue
            weight, bias = _folded_eval_bn(self, fold_bn, 1)
            act, slope = _act_code(fuse_act)
            return ops.ConvTranspose2dFn.apply(x, weight, bias, self.stride[0], self.padding[0], self.output_padding[0], act, slope)
        _to_channels_last_(self)
        weight, bias = self.weight, self.bias
        y = ops.ConvTranspose2dFn.apply(x, weight, bias, self.stride[0], self.padding[0], self.output_padding[0])
        if fuse_act is not None:
            act, slope = _act_code(fuse_act)
            y = ops.ActFn.apply(y, act, slope)
        return y


class _NormMixin:
    def _run(self, x, mode, use_batch_stats, fuse_act, count_batches=True):
        if isinstance(x, Padded):
            raise NotImplementedError('padding in front of a norm layer')
        act, slope = _act_code(fuse_act)
        if use_batch_stats:
            rm = rv = nbt = None
            if self.training and self.track_running_stats:      # BatchNorm2d, and InstanceNorm2d(track_running_stats=True)
                rm, rv = self.running_mean, self.running_var
                if self.momentum is None:
                    raise NotImplementedError('cumulative-average BatchNorm (momentum=None)')
                if count_batches and mode == L.NORM_BATCH:      # bumped by the finalize kernel: no stock torch kernel in the step (torch's
                    # _InstanceNorm.forward never touches num_batches_tracked)
                    nbt = self.num_batches_tracked
            return ops.NormActFn.apply(x, self.weight, self.bias, rm, rv, mode, float(self.eps),
                                       float(self.momentum if self.momentum is not None else 0.0), act, slope, nbt)
        scale, shift = ops.bn_fold(self.weight, self.bias, self.running_mean, self.running_var, float(self.eps))
        return ops.affine_act(x, scale, shift, act, slope)


class BatchNorm2d(_NormMixin, nn.BatchNorm2d):
    def forward(self, x, fuse_act=None, applied=False):
        if applied:      # folded into the preceding conv (eval mode); called only so that forward hooks fire
            return x
        use_batch = self.training or not self.track_running_stats
        return self._run(x, L.NORM_BATCH, use_batch, fuse_act)


class InstanceNorm2d(_NormMixin, nn.InstanceNorm2d):
    def forward(self, x, fuse_act=None, applied=False):
        if applied:      # computed by the producing conv's statistics epilogue; called only so that forward hooks fire
            return x
        # track_running_stats=True (models/networks.py:29-64: --norm instance --norm_track_running_stats): instance statistics + running
        # averages of them in training, the running statistics (a per-channel affine map, like an eval-mode BatchNorm2d) in evaluation
        use_instance = self.training or not self.track_running_stats
        return self._run(x, L.NORM_INSTANCE, use_instance, fuse_act)


_FUSABLE = (Conv2d, ConvTranspose2d, BatchNorm2d, InstanceNorm2d)     # SynchronizedBatchNorm2d is a BatchNorm2d
_NORMS = (BatchNorm2d, InstanceNorm2d)


class FusedSequential(nn.Sequential):
    """nn.Sequential that hands the activation following a conv / norm to that layer's kernel epilogue."""

    def forward(self, x, fuse_act=None):
        """`fuse_act`: an activation module that FOLLOWS this container (nested FusedSequential, e.g. the SPADE discriminator's
        Sequential(Sequential(conv, norm), LeakyReLU)); it is handed to the last layer's epilogue."""
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < n else None
            if isinstance(_inner(x), ops.Normed) and not ((isinstance(m, ReflectionPad2d) and not _hooked(m)) or _edge_conv(m, nxt, x)):
                x = _materialized(x)      # nobody but a quad-granule conv consumes a pending norm (a hooked pad must see a tensor)
            if _edge_conv(m, nxt, x):
                # narrow conv -> train-mode norm -> activation of the generator's edge (inception_generator.py:37-56): the conv runs on the
                # quad-granule kernel with the norm's statistics in its epilogue; in a no-grad forward the normalised tensor is not even
                # written when the next layer can apply scale / shift while staging
                act = mods[i + 2] if i + 2 < n and isinstance(mods[i + 2], _ACTS) else None
                lazy = not torch.is_grad_enabled() and not _hooked(nxt, act) and i + (3 if act is not None else 2) < n
                x = _conv_norm_q(m, nxt, act, x, lazy)
                x = nxt(x, applied=True)
                i += 2
                if act is not None:
                    x = act(x, applied=True)
                    i += 1
                continue
            if isinstance(m, (Conv2d, ConvTranspose2d)) and _bn_folds(nxt):
                # frozen network in eval mode: conv + BatchNorm(running stats) [+ activation] = ONE conv kernel
                act = mods[i + 2] if i + 2 < n and isinstance(mods[i + 2], _ACTS) else None
                if act is None and i + 2 == n:
                    act, tail = fuse_act, True
                else:
                    tail = False
                x = m(x, fuse_act=act, fold_bn=nxt)
                x = nxt(x, applied=True)
                i += 2
                if tail:
                    fuse_act = None if act is not None else fuse_act
                elif act is not None:
                    x = act(x, applied=True)
                    i += 1
            elif isinstance(m, _FUSABLE + (FusedSequential,)) and isinstance(nxt, _ACTS):
                x = m(x, fuse_act=nxt)
                x = nxt(x, applied=True)
                i += 2
            elif i + 1 == n and fuse_act is not None and isinstance(m, _FUSABLE + (FusedSequential,)):
                x = m(x, fuse_act=fuse_act)
                fuse_act = None
                i += 1
            else:
                x = m(x)
                i += 1
        x = _materialized(x)
        if fuse_act is not None:         # nothing could absorb it
            x = fuse_act(x)
        return x


# ---------------------------------------------------------------------------------------------- the generator's edge on the quad-granule kernel
_QCONV = os.environ.get('CAT_QCONV', '1') != '0'      # A/B switch
_QCONV_MAX_C = int(os.environ.get('CAT_QCONV_MAX_C', '80'))


def _inner(x):
    return x.x if isinstance(x, Padded) else x


def _materialized(x):
    if isinstance(x, Padded):
        return Padded(_materialized(x.x), x.pad, x.mode) if isinstance(x.x, ops.Normed) else x
    return x.materialize() if isinstance(x, ops.Normed) else x


def _hooked(*mods):
    return any(m is not None and (m._forward_hooks or m._forward_pre_hooks) for m in mods)


def _edge_conv(m, nxt, x):
    """conv `m` followed by norm `nxt` takes the quad-granule path (csrc/conv_q.hip: cat_qconv_fwd with statistics): dense Conv2d with few
    channels on both sides on a large plane -- the image stem and the first stride-2 conv of a pruned student (3 -> 25 7x7, 25 -> 40 3x3 / 2:
    measured 175 + 105 us against 223 + 108 us for the im2col kernels + 30 us of separate statistics passes, profiles/r04_qconv_layers.txt)"""
    if not _QCONV or not isinstance(m, Conv2d) or not isinstance(nxt, _NORMS) or isinstance(nxt, SynchronizedBatchNorm2d):
        return False
    if _hooked(m):      # the edge path never calls m.forward: a forward (pre-)hook registered on the conv would silently stop firing
        return False
    if m.groups != 1 or 'weight_orig' in m._parameters or m.dilation != (1, 1) or m.padding_mode != 'zeros' or m.stride[0] != m.stride[1]:
        return False
    if isinstance(nxt, BatchNorm2d) and not (nxt.training or not nxt.track_running_stats):
        return False      # running statistics: the folding path
    if isinstance(nxt, BatchNorm2d) and nxt.training and nxt.track_running_stats and nxt.momentum is None:
        return False
    if isinstance(nxt, InstanceNorm2d) and nxt.track_running_stats:
        return False
    inner = _inner(x)
    t = inner.z if isinstance(inner, ops.Normed) else inner
    if not torch.is_tensor(t) or not t.is_cuda or t.dim() != 4:
        return False
    pad = x.pad if isinstance(x, Padded) else m.padding[0]
    if isinstance(x, Padded) and (m.padding[0] != 0 or x.mode not in (L.PAD_ZERO, L.PAD_REFLECT)):
        return False
    from . import qconv
    if not qconv.Layer.supported('conv', m.weight, m.stride[0], pad):
        return False
    n, c, h, w = t.shape
    k, st = m.kernel_size[0], m.stride[0]
    if max(m.in_channels, m.out_channels) > _QCONV_MAX_C or c != m.in_channels:
        return False
    if pad >= min(h, w):
        return False
    ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    if n * ((ho + 7) // 8) * ((wo + 15) // 16) < ops._TCONV_MIN_TILES:
        return False
    try:      # a geometry the kernel has no plan for (cat_qconv_plan fails) declines HERE: the layer takes the general path
        _q_layer(m, pad, x.mode if isinstance(x, Padded) else L.PAD_ZERO).plan_for(t)
    except RuntimeError:
        return False
    return True


def _q_layer(conv, pad, mode):
    from . import qconv
    _to_channels_last_(conv)
    layer = getattr(conv, '_cat_q', None)
    if layer is None or layer.weight is not conv.weight or layer.pad != pad or layer.reflect != (mode == L.PAD_REFLECT):
        layer = conv._cat_q = qconv.Layer('conv', conv.weight, stride=conv.stride[0], pad=pad, reflect=mode == L.PAD_REFLECT)
    return layer


def _conv_norm_q(conv, norm, act_mod, x, lazy):
    from . import qconv
    pad, mode = conv.padding[0], L.PAD_ZERO
    if isinstance(x, Padded):
        x, pad, mode = x.x, x.pad, x.mode
    layer = _q_layer(conv, pad, mode)
    act, slope = _act_code(act_mod)
    inst = isinstance(norm, InstanceNorm2d)
    nmode = L.NORM_INSTANCE if inst else L.NORM_BATCH
    track = (not inst) and norm.training and norm.track_running_stats
    rm, rv, nbt = (norm.running_mean, norm.running_var, norm.num_batches_tracked) if track else (None, None, None)
    eps, mom = float(norm.eps), float(norm.momentum if norm.momentum is not None else 0.0)
    q = {'layer': layer, 'stats': True}
    if isinstance(x, ops.Normed) or lazy:      # no-grad forward
        pre = x.pre() if isinstance(x, ops.Normed) else None
        src = x.z if isinstance(x, ops.Normed) else ops.conform(x)
        n, c, h, w = src.shape
        cout, ho, wo = layer.out_shape(src)
        z = ops.empty_act(n, cout, ho, wo, src.device)
        ops.run_qconv(q, src, conv.bias, z, pre=pre)
        tiles = dict(table=q['table'], plan=q['plan'], scs=q['scs'], lat=(ho, wo), ncls=1)
        groups = n if inst else 1
        scale, shift = ops.norm_from_tiles(tiles, n, cout, groups, norm.weight, norm.bias, rm, rv, nbt, eps, mom)
        out = ops.Normed(z, scale, shift, groups, act, slope)
        return out if lazy else out.materialize()
    z = ops.Conv2dFn.apply(x, conv.weight, conv.bias, conv.stride[0], pad, mode, L.ACT_NONE, 0.0, q)
    tiles = dict(table=q['table'], plan=q['plan'], scs=q['scs'], lat=(z.shape[2], z.shape[3]), ncls=1)
    return ops.NormActFn.apply(z, norm.weight, norm.bias, rm, rv, nmode, eps, mom, act, slope, nbt, tiles)


# ---------------------------------------------------------------------------------------------- SPADE / GauGAN layers
class SynchronizedBatchNorm2d(BatchNorm2d):
    """models/modules/sync_batchnorm/batchnorm.py:SynchronizedBatchNorm2d.  Training mode: statistics over the samples of ALL
    ranks (one RCCL all-reduce of [sum x | sum x^2] per layer, installed with ops.set_bn_sync); with one rank it is
    F.batch_norm, exactly as the reference falls back (:69-72).  `num_batches_tracked` is never advanced (the reference's
    forward bypasses _BatchNorm.forward)."""

    def forward(self, x, fuse_act=None, applied=False):
        if applied:
            return x
        if isinstance(x, Padded):
            raise NotImplementedError('padding in front of a norm layer')
        if self.training or not self.track_running_stats:
            if ops.bn_sync() is None:     # one rank: F.batch_norm (batchnorm.py:69-72) = the fused stats/finalize/apply launch
                return self._run(x, L.NORM_BATCH, True, fuse_act, count_batches=False)
            act, slope = _act_code(fuse_act)
            track = self.training and self.track_running_stats
            return ops.SyncBNFn.apply(x, self.weight, self.bias, self.running_mean if track else None,
                                      self.running_var if track else None, float(self.eps), float(self.momentum), act, slope)
        return self._run(x, L.NORM_BATCH, False, fuse_act)


class Upsample(nn.Upsample):
    """nn.Upsample(scale_factor=k), nearest (inception_spade_generator.py:45)."""

    def forward(self, x):
        if self.mode != 'nearest' or self.size is not None:
            raise NotImplementedError('Upsample: nearest with an integer scale_factor only')
        f = int(self.scale_factor)
        if f != self.scale_factor:
            raise NotImplementedError('Upsample: integer scale_factor only')
        return ops.interp_nearest(x, (x.shape[2] * f, x.shape[3] * f))


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if self.kernel_size not in (2, (2, 2)) or self.stride not in (2, (2, 2)) or self.padding not in (0, (0, 0)):
            raise NotImplementedError('MaxPool2d: only kernel 2, stride 2, padding 0 (VGG19)')
        return ops.MaxPool2x2Fn.apply(x)


def spectral_norm(module, name='weight', n_power_iterations=1, eps=1e-12):
    """torch.nn.utils.spectral_norm for a cat_amd Conv2d: same parameter / buffer names (`weight_orig`, `weight_u`, `weight_v`), same
    initialisation of u and v, one power iteration per training-mode forward.  As in torch, `module.weight` stays behind as a
    plain tensor (which is what init_weights then re-initialises, not weight_orig)."""
    if not isinstance(module, Conv2d) or module.groups != 1 or name != 'weight' or n_power_iterations != 1:
        raise NotImplementedError('spectral_norm: dense cat_amd.nn.Conv2d weights, one power iteration')
    weight = module._parameters.pop('weight')
    o = weight.shape[0]
    k = weight[0].numel()
    with torch.no_grad():
        u = torch.nn.functional.normalize(weight.new_empty(o).normal_(0, 1), dim=0, eps=eps)
        v = torch.nn.functional.normalize(weight.new_empty(k).normal_(0, 1), dim=0, eps=eps)
    module.register_parameter('weight_orig', weight)
    setattr(module, 'weight', weight.data.clone())
    module.register_buffer('weight_u', u)
    module.register_buffer('weight_v', v)
    module._cat_sn_eps = eps
    return module


def remove_spectral_norm(module, name='weight'):
    """torch.nn.utils.remove_spectral_norm for a cat_amd Conv2d (the export path, reference inception_modules.py:314-315,341-342): bake
    weight = weight_orig / sigma with the CURRENT u, v (no power iteration, as SpectralNorm.remove does) into a plain `weight` parameter
    and drop `weight_orig` / `weight_u` / `weight_v`.  One-time host-driven arithmetic, not part of the step."""
    if name != 'weight' or 'weight_orig' not in module._parameters:
        raise ValueError("spectral_norm of '{}' not found in {}".format(name, module))
    with torch.no_grad():
        w = module.weight_orig.detach()
        sigma = torch.dot(module.weight_u, torch.mv(w.reshape(w.shape[0], -1), module.weight_v))
        baked = w / sigma
        if w.is_cuda and w.dim() == 4 and w.shape[1] > 1:      # keep the kernels' padded channels_last storage
            new = ops.padded_weight_like(w.shape, w.device)
            new.copy_(baked)
            baked = new
    if 'weight' in module.__dict__:
        del module.__dict__['weight']
    del module._parameters['weight_orig']
    del module._buffers['weight_u']
    del module._buffers['weight_v']
    module.__dict__.pop('_cat_sn_eps', None)
    module.register_parameter('weight', nn.Parameter(baked))
    return module


def _sn_weight(module):
    w = module.weight_orig
    if w.is_cuda and ops.weight_wcs(w) != ops.cs_for(w.shape[1]):
        _rehouse_(w)
    return ops.SpectralNormFn.apply(w, module.weight_u, module.weight_v, module.training, module._cat_sn_eps)
