"""load_pretrained_weight for `inception_9blocks` generators (reference utils/weight_transfer.py:8-135, 214-266): initialise a
narrower student from a wider pretrained network by keeping, layer by layer, the input / output channels with the largest L1 filter
mass.  Host-side, runs once before training; the SELECTION (torch.topk over |w| sums, fp32, CPU) is what has to match the reference
bit for bit (tests/test_prune.py against tests/golden/weight_transfer.npz).

The `inception_spade` variant (weight_transfer.py:137-212, 267-288; called by SPADEDistillerModules.load_networks when
`restore_pretrained_G_path` is set) is reproduced AS THE REFERENCE BEHAVES, which is narrower than its intent: the SPADE-block rule tests
`isinstance(layer, ConvBNReLU)` / `nn.Conv2d` against layers that are `ConvSyncBNReLU` / `Conv` wrappers, so the main-branch convolutions
of a block are never transferred (only its SPADE norm nets and the learned shortcut are), and the gamma|beta convolutions of an
InceptionSPADE receive the block's INPUT index as output index -- the student's [2C, m, k, k] weight is replaced by the C selected
gamma rows.  The GauGAN launch scripts do not set the flag (the student comes from `shrink_spade_model`); the branch exists so that a
checkpoint prepared with the reference's tool chain loads identically, and tests/test_prune.py pins it to the reference's own run."""
import torch
from torch import nn

from .inception_modules import Conv, ConvBNReLU, ConvSyncBNReLU, InceptionSPADE, InvertedResidualChannels, SPADEInvertedResidualChannels

_NORMS = (nn.InstanceNorm2d, nn.BatchNorm2d)


def _set(param, value):
    param.data = value.clone().to(param.device)


def _topk(q, k):
    return q.topk(k, largest=True)[1]


def transfer_conv2d(m1, m2, input_index=None, output_index=None):
    dw = m1.in_channels == m1.groups and m1.groups > 1
    w1 = m1.weight.data.detach().float().cpu().contiguous()
    if m1.out_channels == 3:                      # image head: keep all outputs, select inputs
        assert input_index is not None
        _set(m2.weight, w1[input_index] if dw else w1[:, input_index])
        if m2.bias is not None:
            _set(m2.bias, m1.bias.data.cpu())
        return None
    if m1.in_channels == 3:
        assert input_index is None
        input_index = [0, 1, 2]
    p = w1
    if input_index is None:
        idxs = _topk(p.abs().sum([0, 2, 3]), m2.in_channels)
        p = p[idxs] if dw else p[:, idxs]
    else:
        p = p[input_index] if dw else p[:, input_index]
    idxs = _topk(p.abs().sum([1, 2, 3]), m2.out_channels) if output_index is None else output_index
    _set(m2.weight, p[idxs])
    if m2.bias is not None:
        _set(m2.bias, m1.bias.data.cpu()[idxs])
    return idxs


def transfer_conv_transpose2d(m1, m2, input_index=None, output_index=None):
    assert output_index is None
    dw = m1.out_channels == m1.groups and m1.groups > 1
    p = m1.weight.data.detach().float().cpu().contiguous()
    p = p[_topk(p.abs().sum([1, 2, 3]), m2.in_channels)] if input_index is None else p[input_index]
    idxs = _topk(p.abs().sum([0, 2, 3]), m2.out_channels)
    _set(m2.weight, p[idxs] if dw else p[:, idxs])
    if m2.bias is not None:
        _set(m2.bias, m1.bias.data.cpu()[idxs])
    return idxs


def transfer_norm(m1, m2, input_index=None, output_index=None):
    assert type(m1) == type(m2)
    if m1.weight is not None and m2.weight is not None:
        _set(m2.weight, m1.weight.data.cpu()[input_index])
    if m1.bias is not None and m2.bias is not None:
        _set(m2.bias, m1.bias.data.cpu()[input_index])
    return input_index        # running statistics are NOT transferred (weight_transfer.py:83-94)


def transfer_block(m1, m2, input_index=None, output_index=None):
    assert output_index is None
    idxs = input_index
    for ops1, ops2 in ((m1.res_ops, m2.res_ops), (m1.dw_ops, m2.dw_ops)):
        for op1, op2 in zip(ops1, ops2):
            idxs = input_index
            for layer1, layer2 in zip(op1, op2):
                assert type(layer1) == type(layer2)
                if isinstance(layer1, ConvBNReLU):
                    idxs = transfer(layer1, layer2, input_index=idxs)
                if isinstance(layer2, nn.Conv2d):
                    idxs = transfer(layer1, layer2, input_index=idxs, output_index=input_index)
    return transfer(m1.pw_bn, m2.pw_bn, input_index=idxs)


def transfer_spade_block(m1, m2, input_index=None, output_index=None):
    """weight_transfer.py:150-184.  The two isinstance tests below can never hold for a SPADE block's layers (ConvSyncBNReLU / Conv
    wrappers) -- kept literally: the main-branch convolutions stay as they are."""
    assert output_index is None
    idxs_first = transfer(m1.spade, m2.spade, input_index=input_index)
    idxs = idxs_first
    for ops1, ops2 in ((m1.res_ops, m2.res_ops), (m1.dw_ops, m2.dw_ops)):
        for op1, op2 in zip(ops1, ops2):
            idxs = idxs_first
            for layer1, layer2 in zip(op1, op2):
                assert type(layer1) == type(layer2)
                if isinstance(layer1, ConvBNReLU):
                    idxs = transfer(layer1, layer2, input_index=idxs)
                if isinstance(layer2, nn.Conv2d):
                    idxs = transfer(layer1, layer2, input_index=idxs)
    if m1.shortcut is not None:
        assert m2.shortcut is not None
        idxs = transfer(m1.shortcut[0], m2.shortcut[0], input_index=input_index)
        idxs = transfer(m1.shortcut[1], m2.shortcut[1], input_index=idxs)
    else:
        assert m2.shortcut is None
    return idxs


def transfer_inception_spade(m1, m2, input_index=None, output_index=None):
    """weight_transfer.py:187-212: first convs of the gamma|beta nets keep every segmentation channel and the top-k hidden channels;
    the closing nn.Conv2d gets (hidden index, the normalised tensor's channel index) -- i.e. the gamma rows only (module docstring)."""
    idxs = transfer(m1.param_free_norm, m2.param_free_norm, input_index=input_index)
    for ops1, ops2 in ((m1.res_ops, m2.res_ops), (m1.dw_ops, m2.dw_ops)):
        for op1, op2 in zip(ops1, ops2):
            for layer1, layer2 in zip(op1, op2):
                assert type(layer1) == type(layer2)
                if isinstance(layer1, ConvSyncBNReLU):
                    idxs = transfer(layer1, layer2, input_index=list(range(layer1.conv.in_channels)))
                if isinstance(layer1, nn.Conv2d):
                    idxs = transfer(layer1, layer2, idxs, input_index)
    return input_index


def transfer(m1, m2, input_index=None, output_index=None):
    if isinstance(m1, ConvSyncBNReLU):      # weight_transfer.py:137-141
        idxs = transfer(m1.conv, m2.conv, input_index=input_index)
        return transfer(m1.norm, m2.norm, input_index=idxs)
    if isinstance(m1, Conv):                # :144-147
        return transfer(m1.conv, m2.conv, input_index=input_index)
    if isinstance(m1, InceptionSPADE):
        return transfer_inception_spade(m1, m2, input_index, output_index)
    if isinstance(m1, SPADEInvertedResidualChannels):
        return transfer_spade_block(m1, m2, input_index, output_index)
    if isinstance(m1, ConvBNReLU):
        idxs = transfer(m1[0], m2[0], input_index=input_index)
        return transfer(m1[1], m2[1], input_index=idxs)
    if isinstance(m1, nn.Conv2d):
        return transfer_conv2d(m1, m2, input_index, output_index)
    if isinstance(m1, nn.ConvTranspose2d):
        return transfer_conv_transpose2d(m1, m2, input_index, output_index)
    if isinstance(m1, _NORMS):
        return transfer_norm(m1, m2, input_index, output_index)
    if isinstance(m1, InvertedResidualChannels):
        return transfer_block(m1, m2, input_index, output_index)
    raise NotImplementedError('Unknown module [%s]!' % type(m1))


def load_pretrained_weight(model1, model2, netA, netB, ngf1, ngf2):
    """reference utils/weight_transfer.py:240-288."""
    assert ngf1 >= ngf2
    if model1 == 'inception_spade':
        with torch.no_grad():
            idxs = transfer(netA.fc, netB.fc, list(range(netA.fc.in_channels)))
            idxs = transfer(netA.fc_norm, netB.fc_norm, idxs)
            for name in ('head_0', 'G_middle_0', 'G_middle_1', 'up_0', 'up_1', 'up_2', 'up_3'):
                idxs = transfer(getattr(netA, name), getattr(netB, name), idxs)
            if hasattr(netA, 'up_4'):
                assert hasattr(netB, 'up_4')
                idxs = transfer(netA.up_4, netB.up_4, idxs)
            else:
                assert not hasattr(netB, 'up_4')
            transfer(netA.conv_img, netB.conv_img, idxs)
        return idxs
    if model1 != 'inception_9blocks':
        raise NotImplementedError('Unknown model [%s]!' % model1)
    kinds = (nn.Conv2d, nn.ConvTranspose2d) + _NORMS
    index = None
    with torch.no_grad():
        for part, extra in (('down_sampling', ()), ('features', (InvertedResidualChannels,)), ('up_sampling', ())):
            s1, s2 = getattr(netA, part), getattr(netB, part)
            assert len(s1) == len(s2)
            for m1, m2 in zip(s1, s2):
                if isinstance(m1, kinds + extra):
                    index = transfer(m1, m2, index)
    return index
