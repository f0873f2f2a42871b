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
from .inception_modules import (Conv, ConvSyncBNReLU, InceptionSPADE, InvertedResidualChannels, SPADEInvertedResidualChannels)
from .optim import FusedAdam


def get_bn_to_prune(model, verbose=False, spade=False):
    """utils/prune.py:5-61: ordered names of the first-BN scales of every block (SPADE: the block's own branches, then the
    branches of its InceptionSPADE)."""
    weights = []
    for name, m in model.get_named_block_list().items():
        if spade:
            if isinstance(m, SPADEInvertedResidualChannels):
                weights += ['{}.weight'.format(n) for n in m.get_named_first_res_bn(prefix=name)]
                weights += ['{}.weight'.format(n) for n in m.get_named_first_dw_bn(prefix=name)]
                weights += ['{}.weight'.format(n) for n in m.spade.get_named_first_res_bn(prefix=name + '.spade')]
                weights += ['{}.weight'.format(n) for n in m.spade.get_named_first_dw_bn(prefix=name + '.spade')]
        elif isinstance(m, InvertedResidualChannels):
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
    if isinstance(m, ConvSyncBNReLU):          # model_profiling.py:179-185
        shape = _profile(m.conv, shape)
        shape = _profile(m.norm, shape)
        m.n_macs = m.conv.n_macs + m.norm.n_macs
        return shape
    if isinstance(m, Conv):                    # :186-191
        shape = _profile(m.conv, shape)
        m.n_macs = m.conv.n_macs
        return shape
    if isinstance(m, InceptionSPADE):          # :192-201; the branches run on the (resized) segmentation map
        seg_shape = (n, m.input_dim, h, w)
        _profile(m.param_free_norm, shape)
        m.n_macs = m.param_free_norm.n_macs
        for op in list(m.res_ops) + list(m.dw_ops):
            _profile(op, seg_shape)
            m.n_macs += op.n_macs
        return shape
    if isinstance(m, SPADEInvertedResidualChannels):   # :166-178
        m.n_macs = 0
        out_shape = (n, m.output_dim, h, w)
        if len(m.res_ops) + len(m.dw_ops) == 0:       # forward returns before the SPADE / branch hooks can fire
            if m.shortcut is not None:
                _profile(m.shortcut, shape)
                m.n_macs += m.shortcut.n_macs
            return out_shape
        _profile(m.spade, shape)
        for op in list(m.res_ops) + list(m.dw_ops):
            _profile(op, shape)
            m.n_macs += op.n_macs
        if m.shortcut is not None:
            _profile(m.shortcut, shape)
            m.n_macs += m.shortcut.n_macs
        m.n_macs += m.spade.n_macs
        return out_shape
    m.n_macs = 0
    return shape


def _profile_spade_generator(model, shape):
    """InceptionSPADEGenerator.forward (inception_spade_generator.py:63-124) as shape propagation: n_macs = sum over the direct
    children (model_profiling.py:203-213); nn.Upsample and activations count zero."""
    n, c, h, w = shape
    cur = _profile(model.fc, (n, c, model.sh, model.sw))
    cur = _profile(model.fc_norm, cur)
    total = model.fc.n_macs + model.fc_norm.n_macs

    def up(sh):
        return (sh[0], sh[1], sh[2] * 2, sh[3] * 2)

    nul = model.opt.num_upsampling_layers
    cur = _profile(model.head_0, cur)
    cur = _profile(model.G_middle_0, up(cur))
    cur = _profile(model.G_middle_1, up(cur) if nul in ('more', 'most') else cur)
    total += model.head_0.n_macs + model.G_middle_0.n_macs + model.G_middle_1.n_macs
    for name in ('up_0', 'up_1', 'up_2', 'up_3') + (('up_4',) if nul == 'most' else ()):
        blk = getattr(model, name)
        cur = _profile(blk, up(cur))
        total += blk.n_macs
    cur = _profile(model.conv_img, cur)
    total += model.conv_img.n_macs
    model.up.n_macs = 0
    return total


def model_profiling(model, height, width, batch=1, channel=3, **unused):
    """n_macs of an InceptionGenerator (and of .down_sampling / .features / .up_sampling) or of an InceptionSPADEGenerator at
    batch x channel x H x W."""
    shape = (batch, channel, height, width)
    if hasattr(model, 'head_0'):          # InceptionSPADEGenerator
        model.n_macs = _profile_spade_generator(model, shape)
        model.n_params = sum(p.numel() for p in model.parameters())
        return model.n_macs, model.n_params
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


def spade_search(teacher, target_flops, opt):
    """The threshold search of shrink_spade_model (utils/common.py:710-812): binary search on |gamma| of fc_norm + every first BN of
    the blocks and of their SPADE layers; the trunk width is count(|fc_norm.gamma| > thr) rounded DOWN to a multiple of 16 (32 for
    'most') with the prune_cin_lb / prune_cin_ub floors applied in units of that multiple.  Returns
    (threshold, searched n_macs, student network with the searched ARCHITECTURE and freshly initialised weights)."""
    netG_tmp = copy.deepcopy(teacher).cpu()
    named = dict(netG_tmp.named_parameters())
    weights = [netG_tmp.fc_norm.weight] + [named[n] for n in get_bn_to_prune(netG_tmp, spade=True)]
    allw = torch.cat([w.detach().abs().float() for w in weights])
    lb, ub = allw.min(), allw.max()
    searched, thr, cand = float('inf'), None, None
    nul = opt.num_upsampling_layers
    ch_div = 32 if nul == 'most' else 16
    features = ['head_0', 'G_middle_0', 'G_middle_1'] + ['up_%d' % i for i in range(5 if nul == 'most' else 4)]
    norm_layer = type(netG_tmp.fc_norm)
    while (abs(ub - lb) > 1e-3 * lb) or (searched > target_flops):
        cand = copy.deepcopy(netG_tmp)
        thr = (lb + ub) / 2
        out_channels = (cand.fc_norm.weight.detach().abs() > thr).sum().item()
        out_channels = max(out_channels // ch_div, getattr(opt, 'prune_cin_lb', 1)) * ch_div
        out_channels = min(out_channels // ch_div, getattr(opt, 'prune_cin_ub', float('inf'))) * ch_div
        ngf_stu = out_channels // 16
        cand.fc_norm = norm_layer(out_channels, affine=True)
        old = cand.fc
        cand.fc = cnn.Conv2d(old.in_channels, out_channels, kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                             bias=old.bias is not None)
        in_channels = out_channels
        for layer_name in features:
            layer = getattr(cand, layer_name)
            layer.input_dim = in_channels
            out_channels = in_channels // 2 if 'up' in layer_name else in_channels
            layer.output_dim = out_channels
            layer.res_channels = [int((bn.weight.detach().abs() > thr).sum().item()) for bn in layer.get_first_res_bn()]
            layer.dw_channels = [int((bn.weight.detach().abs() > thr).sum().item()) for bn in layer.get_first_dw_bn()]
            layer.spade.output_dim = layer.input_dim
            layer.spade.res_channels = [int((bn.weight.detach().abs() > thr).sum().item()) for bn in layer.spade.get_first_res_bn()]
            layer.spade.dw_channels = [int((bn.weight.detach().abs() > thr).sum().item()) for bn in layer.spade.get_first_dw_bn()]
            layer.res_ops, layer.dw_ops, layer.shortcut, layer.spade = layer._build(build_only=True)
            in_channels = out_channels
        old = cand.conv_img
        cand.conv_img = cnn.Conv2d(in_channels, old.out_channels, kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                                   bias=old.bias is not None)
        searched, _ = model_profiling(cand, opt.data_height, opt.data_width, channel=opt.data_channel)
        if searched > target_flops:
            lb = thr
        else:
            ub = thr
    return thr, searched, cand, ngf_stu


def shrink_spade_model(model, target_flops, opt, verbose=True):
    """utils/common.py:710-869: the student becomes the searched architecture (no weight copy: its tensors keep torch's default
    initialisation, as in the reference), netAs / optimizer_G / schedulers are rebuilt."""
    from . import networks
    m = model.modules_on_one_gpu
    thr, searched, student, ngf_stu = spade_search(m.netG_teacher, target_flops, opt)
    if verbose:
        print(f'scale threshold: {thr}, searched flops: {searched}, target flops: {target_flops}, '
              f'flops diff: {searched - target_flops}.')
    m.netG_student = student.to(model.device)
    model.shrink_threshold = thr
    model_profiling(m.netG_student, opt.data_height, opt.data_width, channel=opt.data_channel)
    netAs = nn.ModuleList()
    for mapping_layer in m.mapping_layers:
        if mapping_layer != 'up_1':
            fs, ft = ngf_stu * 16, opt.teacher_ngf * 16
        else:
            fs, ft = ngf_stu * 4, opt.teacher_ngf * 4
        netAs.append(cnn.Conv2d(in_channels=fs, out_channels=ft, kernel_size=1))
    m.netAs = netAs.to(model.device)
    if opt.no_TTUR:
        beta1, beta2, g_lr = opt.beta1, opt.beta2, opt.lr
    else:
        beta1, beta2, g_lr = 0.0, 0.9, opt.lr / 2
    g_params = list(m.netG_student.parameters())
    for netA in m.netAs:
        g_params += list(netA.parameters())
    model.optimizer_G = FusedAdam(g_params, lr=g_lr, betas=(beta1, beta2))
    model.optimizers = [model.optimizer_G, model.optimizer_D]
    if model.isTrain:
        model.schedulers = [networks.get_scheduler(o, opt) for o in model.optimizers]
    if verbose:
        print('All layers are pruned.')
    return thr, searched


def shrink(model, opt):
    """reference utils/common.py:872-878."""
    target_flops = getattr(opt, 'target_flops', 0.0)
    assert target_flops > 0
    if 'spade' in getattr(opt, 'distiller', 'inception'):
        return shrink_spade_model(model, target_flops, opt)
    return shrink_model(model, target_flops, opt)
