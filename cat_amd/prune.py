"""Host-side structure search that defines every student shape before the hot loop starts.

Mirrors, on cat_amd modules, what the reference does once per run on the CPU/GPU with stock torch:
  get_bn_to_prune  utils/prune.py:5-61          ordered names of the norm scales that drive pruning
  model_profiling  utils/model_profiling.py:65-135,269-346   the MAC counter `n_macs` (here: shape propagation, no forward pass)
  shrink_model     utils/common.py:315-707      binary search of the |gamma| threshold to a MAC budget, then a masked
                                                weight copy teacher -> student
The results are integers / boolean masks and are required to be bit-exact with the reference (tests/test_prune.py
against tests/golden/shrink_*.npz).  All float comparisons are float32 tensor ops, exactly as the reference performs them.
"""
import copy
import itertools

import torch
from torch import nn

from . import nn as cnn
from .inception_modules import InvertedResidualChannels
from .optim import FusedAdam


def get_bn_to_prune(model, verbose=False):
    weights = []
    for name, m in model.get_named_block_list().items():
        if isinstance(m, InvertedResidualChannels):
            weights += ['{}.weight'.format(n) for n in m.get_named_first_res_bn(prefix=name)]
            weights += ['{}.weight'.format(n) for n in m.get_named_first_dw_bn(prefix=name)]
    if verbose:
        for n in weights:
            print(n)
    keys = set(k for k, _ in model.named_parameters())
    for n in weights:
        assert n in keys
    return weights


# ---------------------------------------------------------------------------------------------- MAC counter
def _profile(m, shape):
    """Set m.n_macs (+ on all descendants) for an input of `shape` = (N, C, H, W); return the output shape."""
    n, c, h, w = shape
    if isinstance(m, nn.Conv2d):
        kh, kw = m.kernel_size
        ho = (h + 2 * m.padding[0] - kh) // m.stride[0] + 1
        wo = (w + 2 * m.padding[1] - kw) // m.stride[1] + 1
        m.n_macs = (c * m.out_channels * kh * kw * ho * wo // m.groups) * n
        return (n, m.out_channels, ho, wo)
    if isinstance(m, nn.ConvTranspose2d):
        kh, kw = m.kernel_size
        ho = (h - 1) * m.stride[0] - 2 * m.padding[0] + kh + m.output_padding[0]
        wo = (w - 1) * m.stride[1] - 2 * m.padding[1] + kw + m.output_padding[1]
        m.n_macs = (c * m.out_channels * kh * kw * ho * wo // m.groups) * n
        return (n, m.out_channels, ho, wo)
    if isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
        m.n_macs = 0 if m.track_running_stats else c * h * w * n
        return shape
    if isinstance(m, (nn.ReflectionPad2d, nn.ConstantPad2d)):
        m.n_macs = 0
        p = m.padding
        return (n, c, h + p[2] + p[3], w + p[0] + p[1])
    if isinstance(m, InvertedResidualChannels):
        m.n_macs = 0
        if len(m.res_ops) + len(m.dw_ops) == 0:
            return shape
        for op in list(m.res_ops) + list(m.dw_ops):
            _profile(op, shape)
            m.n_macs += op.n_macs
        _profile(m.pw_bn, shape)
        m.n_macs += m.pw_bn.n_macs
        return shape
    if isinstance(m, nn.Sequential):
        m.n_macs = 0
        for sub in m:
            shape = _profile(sub, shape)
            m.n_macs += getattr(sub, 'n_macs', 0)
        return shape
    m.n_macs = 0
    return shape


def model_profiling(model, height, width, batch=1, channel=3, **unused):
    """n_macs of an InceptionGenerator (and of .down_sampling / .features / .up_sampling) at batch x channel x H x W."""
    shape = (batch, channel, height, width)
    total = 0
    for part in (model.down_sampling, model.features, model.up_sampling):
        shape = _profile(part, shape)
        total += part.n_macs
    model.n_macs = total
    model.n_params = sum(p.numel() for p in model.parameters())
    return model.n_macs, model.n_params


# ---------------------------------------------------------------------------------------------- shrink
def _norm_cls(opt):
    return {'instance': cnn.InstanceNorm2d, 'batch': cnn.BatchNorm2d}[opt.norm]


def _new_norm(opt, channels):
    return _norm_cls(opt)(channels, affine=opt.norm_affine, track_running_stats=opt.norm_track_running_stats)


def _clone_conv(old, cin, cout, transposed=False):
    kw = dict(kernel_size=old.kernel_size, stride=old.stride, padding=old.padding, bias=old.bias is not None)
    if transposed:
        return cnn.ConvTranspose2d(cin, cout, output_padding=old.output_padding, **kw)
    return cnn.Conv2d(cin, cout, **kw)


def _kept(norm, thr):
    return norm.weight.detach().abs() > thr


def _mask_with_floor(norm, thr, floor):
    """|gamma| > thr, or -- if fewer than `floor` channels survive -- the `floor` largest (ties included)."""
    g = norm.weight.detach().abs()
    m = g > thr
    if m.sum().item() < floor:
        private = torch.sort(g.view(-1), descending=True)[0][floor - 1]
        m = g >= private
    return m


def _copy_norm(dst, src, mask):
    dst.weight.data.copy_(src.weight.data[mask])
    dst.bias.data.copy_(src.bias.data[mask])
    if src.track_running_stats:
        assert dst.track_running_stats
        dst.running_mean.data.copy_(src.running_mean.data[mask])
        dst.running_var.data.copy_(src.running_var.data[mask])
        dst.num_batches_tracked.data.copy_(src.num_batches_tracked.data)


def _norm_indices(seq, cls):
    return [i for i, layer in enumerate(seq) if isinstance(layer, cls)]


def _apply_structure(net, src, thr, opt, copy_weights):
    """Rebuild `net` (a deepcopy of the teacher `src`) at threshold `thr`.  With copy_weights the surviving teacher
    weights are copied in (the final pass, utils/common.py:445-662); without, only shapes matter (the search pass)."""
    cls = _norm_cls(opt)
    lb = getattr(opt, 'prune_cin_lb', 1)
    ub = getattr(opt, 'prune_cin_ub', float('inf'))
    ft_lb = getattr(opt, 'prune_ft_cin_lb', 1)
    ds_idx, us_idx = _norm_indices(net.down_sampling, cls), _norm_indices(net.up_sampling, cls)
    in_ch, in_mask = None, None
    masks = {'down': [], 'blocks': [], 'up': []}

    for idx in ds_idx:
        old_norm, old_conv = src.down_sampling[idx], src.down_sampling[idx - 1]
        if copy_weights:
            mask = _mask_with_floor(old_norm, thr, lb)
            if idx == ds_idx[0] and mask.sum().item() > ub:
                private = torch.sort(old_norm.weight.detach().abs().view(-1), descending=False)[0][ub - 1]
                mask = old_norm.weight.detach().abs() <= private
            if idx == ds_idx[-1] and mask.sum().item() < ft_lb:
                mask = _mask_with_floor(old_norm, thr, ft_lb)
            out_ch = int(mask.sum().item())
        else:
            mask = None
            out_ch = max(int(_kept(old_norm, thr).sum().item()), lb)
            if idx == ds_idx[0]:
                out_ch = min(out_ch, ub)
            if idx == ds_idx[-1]:
                out_ch = max(out_ch, ft_lb)
        if in_ch is None:
            in_ch = old_conv.in_channels
        net.down_sampling[idx] = _new_norm(opt, out_ch)
        net.down_sampling[idx - 1] = _clone_conv(old_conv, in_ch, out_ch)
        if copy_weights:
            _copy_norm(net.down_sampling[idx], old_norm, mask)
            w = old_conv.weight.data[mask]
            net.down_sampling[idx - 1].weight.data.copy_(w if in_mask is None else w[:, in_mask])
            if old_conv.bias is not None:
                net.down_sampling[idx - 1].bias.data.copy_(old_conv.bias.data[mask])
            masks['down'].append(mask)
        in_ch, in_mask = out_ch, mask
    trunk = in_ch

    for layer, old in zip(net.features, src.features):
        layer.input_dim = in_ch
        res_m = [_kept(bn, thr) for bn in old.get_first_res_bn()]
        dw_m = [_kept(bn, thr) for bn in old.get_first_dw_bn()]
        layer.res_channels = [int(sum(m).item()) for m in res_m]
        layer.dw_channels = [int(sum(m).item()) for m in dw_m]
        layer.res_ops, layer.dw_ops, layer.pw_bn = layer._build()
        if not copy_weights:
            continue
        masks['blocks'].append((res_m, dw_m))
        j = 0
        for old_op, mid in zip(old.res_ops, res_m):
            if mid.sum() == 0:
                continue
            new_op = layer.res_ops[j]
            new_op[1][0].weight.data.copy_(old_op[1][0].weight.data[mid][:, in_mask])
            if new_op[1][0].bias is not None:
                new_op[1][0].bias.data.copy_(old_op[1][0].bias.data[mid])
            _copy_norm(new_op[1][1], old_op[1][1], mid)
            new_op[4].weight.data.copy_(old_op[4].weight.data[in_mask][:, mid])
            if new_op[4].bias is not None:
                new_op[4].bias.data.copy_(old_op[4].bias.data[in_mask])
            j += 1
        assert len(layer.res_ops) == j
        j = 0
        for old_op, mid in zip(old.dw_ops, dw_m):
            if mid.sum() == 0:
                continue
            new_op = layer.dw_ops[j]
            new_op[0][0].weight.data.copy_(old_op[0][0].weight.data[mid][:, in_mask])
            if new_op[0][0].bias is not None:
                new_op[0][0].bias.data.copy_(old_op[0][0].bias.data[mid])
            _copy_norm(new_op[0][1], old_op[0][1], mid)
            new_op[2][0].weight.data.copy_(old_op[2][0].weight.data[mid])
            if new_op[2][0].bias is not None:
                new_op[2][0].bias.data.copy_(old_op[2][0].bias.data[mid])
            _copy_norm(new_op[2][1], old_op[2][1], mid)
            new_op[4].weight.data.copy_(old_op[4].weight.data[in_mask][:, mid])
            if new_op[4].bias is not None:
                new_op[4].bias.data.copy_(old_op[4].bias.data[in_mask])
            j += 1
        assert len(layer.dw_ops) == j
        # NB: like the reference, pw_bn of the student is freshly initialised (its teacher values are not copied)

    for idx in us_idx:
        old_norm, old_conv = src.up_sampling[idx], src.up_sampling[idx - 1]
        if copy_weights:
            mask = _mask_with_floor(old_norm, thr, lb)
            out_ch = int(mask.sum().item())
        else:
            mask = None
            out_ch = max(int(_kept(old_norm, thr).sum().item()), lb)
        net.up_sampling[idx] = _new_norm(opt, out_ch)
        net.up_sampling[idx - 1] = _clone_conv(old_conv, in_ch, out_ch, transposed=True)
        if copy_weights:
            _copy_norm(net.up_sampling[idx], old_norm, mask)
            net.up_sampling[idx - 1].weight.data.copy_(old_conv.weight.data[in_mask][:, mask])
            if old_conv.bias is not None:
                net.up_sampling[idx - 1].bias.data.copy_(old_conv.bias.data[mask])
            masks['up'].append(mask)
        in_ch, in_mask = out_ch, mask

    last = src.up_sampling[-2]
    net.up_sampling[-2] = _clone_conv(last, in_ch, last.out_channels)
    if copy_weights:
        net.up_sampling[-2].weight.data.copy_(last.weight.data[:, in_mask])
        if last.bias is not None:
            net.up_sampling[-2].bias.data.copy_(last.bias.data)
    return trunk, masks


def search_threshold(teacher, target_flops, opt):
    """Binary search of the scale threshold (utils/common.py:341-443).  Returns (threshold tensor, searched n_macs)."""
    cls = _norm_cls(opt)
    weights = [teacher.down_sampling[i].weight for i in _norm_indices(teacher.down_sampling, cls)]
    named = dict(teacher.named_parameters())
    weights += [named[n] for n in get_bn_to_prune(teacher)]
    weights += [teacher.up_sampling[i].weight for i in _norm_indices(teacher.up_sampling, cls)]
    allw = torch.cat([w.detach().abs().float().cpu() for w in weights])
    lb, ub = allw.min(), allw.max()
    searched, thr = float('inf'), None
    cpu_teacher = copy.deepcopy(teacher).cpu()
    while (abs(ub - lb) > 1e-3 * lb) or (searched > target_flops):
        thr = (lb + ub) / 2
        cand = copy.deepcopy(cpu_teacher)
        _apply_structure(cand, cpu_teacher, thr, opt, copy_weights=False)
        searched, _ = model_profiling(cand, opt.data_height, opt.data_width)
        if searched > target_flops:
            lb = thr
        else:
            ub = thr
    return thr, searched


def shrink_model(model, target_flops, opt, verbose=True):
    """Replace model.netG_student by the teacher pruned to `target_flops`, rebuild netAs / optimizer_G / schedulers and
    re-install the mapping hooks -- the side effects of the reference's shrink_model (utils/common.py:315-707)."""
    from . import networks
    model.remove_mapping_hook()
    teacher = model.netG_teacher
    thr, searched = search_threshold(teacher, target_flops, opt)
    if verbose:
        print(f'scale threshold: {thr}, searched flops: {searched}, target flops: {target_flops}, '
              f'flops diff: {searched - target_flops}.')
    cpu_teacher = copy.deepcopy(teacher).cpu()
    student = copy.deepcopy(cpu_teacher)
    trunk, masks = _apply_structure(student, cpu_teacher, thr, opt, copy_weights=True)
    model.netG_student = student.to(model.device)
    model.shrink_threshold, model.shrink_masks = thr, masks
    model_profiling(model.netG_student, opt.data_height, opt.data_width)

    net_as, g_params = [], []
    for net_a in model.netAs:
        new_a = cnn.Conv2d(in_channels=trunk, out_channels=net_a.out_channels, kernel_size=net_a.kernel_size).to(model.device)
        g_params.append(new_a.parameters())
        net_as.append(new_a)
    model.netAs = net_as
    model.add_mapping_hook()
    model.optimizer_G = FusedAdam([{'params': model.netG_student.parameters()}, {'params': itertools.chain(*g_params)}],
                                  lr=opt.lr, betas=(opt.beta1, 0.999))
    model.optimizers = [model.optimizer_G, model.optimizer_D]
    if model.isTrain:
        model.schedulers = [networks.get_scheduler(o, opt) for o in model.optimizers]
    if verbose:
        print('All layers are pruned.')
    return thr, searched


def shrink(model, opt):
    """reference utils/common.py:872-878 dispatch (inception distillers only in this build)."""
    if 'spade' in getattr(opt, 'distiller', 'inception'):
        raise NotImplementedError('SPADE shrink is outside this round (SURVEY §8f rank 2)')
    return shrink_model(model, opt.target_flops, opt)
