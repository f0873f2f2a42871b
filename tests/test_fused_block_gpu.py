"""GPU: the fused training-mode path of InvertedResidualChannels (cat_amd/fused_block.py) against the general per-layer path of the
same module (itself pinned to the CPU oracle by test_model_gpu.py) and against stock PyTorch on the host."""
import copy

import numpy as np
import pytest
import torch

from oracle import detfill

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _block(norm, dev, inp=77, res=(11, 12, 18), dw=(15, 15, 12), padding='reflect'):
    import functools
    from cat_amd import nn as cnn
    from cat_amd.inception_modules import InvertedResidualChannels, get_active_fn
    nl = cnn.BatchNorm2d if norm == 'batch' else functools.partial(cnn.InstanceNorm2d, affine=False, track_running_stats=False)
    blk = InvertedResidualChannels(inp, list(res), list(dw), 1, [1, 3, 5], [1, 3, 5], padding_type=padding, use_bias=norm != 'batch', norm_layer=nl,
                                   norm_kwargs={'momentum': 0.1, 'eps': 1e-5}, active_fn=get_active_fn('nn.ReLU'))
    sd = blk.state_dict()
    blk.load_state_dict(detfill.fill_state_dict(sd, 77, gamma_abs_normal=True))
    return blk.to(dev).train()


def _torch_twin(blk):
    """The same block out of stock torch.nn layers on the host (the reference's own construction, inception_modules.py:124-180)."""
    from torch import nn
    sd = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items()}
    bn = isinstance(blk.pw_bn, nn.BatchNorm2d)
    mk = (lambda c: nn.BatchNorm2d(c, momentum=0.1, eps=1e-5)) if bn else (lambda c: nn.InstanceNorm2d(c, eps=1e-5))
    pad = nn.ReflectionPad2d if blk.padding_type == 'reflect' else nn.ZeroPad2d
    C = blk.input_dim
    res_ops, dw_ops = nn.ModuleList(), nn.ModuleList()
    for m, k in zip(blk.res_channels, blk.res_kernel_sizes):
        if m:
            res_ops.append(nn.Sequential(pad((k - 1) // 2), nn.Sequential(nn.Conv2d(C, m, k, bias=blk.use_bias), mk(m), nn.ReLU()), nn.Dropout(0.0),
                                         pad((k - 1) // 2), nn.Conv2d(m, C, k, bias=blk.use_bias)))
    for m, k in zip(blk.dw_channels, blk.dw_kernel_sizes):
        if m:
            dw_ops.append(nn.Sequential(nn.Sequential(nn.Conv2d(C, m, 1, bias=blk.use_bias), mk(m), nn.ReLU()), pad((k - 1) // 2),
                                        nn.Sequential(nn.Conv2d(m, m, k, groups=m, bias=blk.use_bias), mk(m), nn.ReLU()), nn.Dropout(0.0),
                                        nn.Conv2d(m, C, 1, bias=blk.use_bias)))
    twin = nn.Module()
    twin.res_ops, twin.dw_ops, twin.pw_bn = res_ops, dw_ops, mk(C)
    twin.load_state_dict(sd)
    twin.train()

    def fwd(x):
        tmp = sum([op(x) for op in twin.res_ops]) + sum([op(x) for op in twin.dw_ops])
        return x + twin.pw_bn(tmp)
    return twin, fwd


@pytest.mark.parametrize('norm,padding,shape', [('batch', 'reflect', (4, 77, 48, 64)), ('instance', 'reflect', (4, 77, 48, 64)),
                                                ('batch', 'zero', (3, 77, 50, 70)), ('batch', 'reflect', (8, 40, 32, 48))])
def test_fused_block_forward_matches_general_path_and_torch(norm, padding, shape):
    from cat_amd import _lib, fused_block, ops
    _lib.load()
    dev = torch.device('cuda:0')
    n, c, h, w = shape
    res, dw = ((11, 12, 18), (15, 15, 12)) if c == 77 else ((7, 0, 9), (16, 5, 0))
    blk = _block(norm, dev, c, res, dw, padding)
    ref_blk = copy.deepcopy(blk)
    x = detfill.normal((n, c, h, w), 5)
    twin, twin_fwd = _torch_twin(blk)
    y_t = twin_fwd(x)
    xg = ops.to_nhwc(x.to(dev))
    with torch.no_grad():
        assert fused_block.applicable(blk, xg)
        y_f = blk(xg)
        fused_block.set_enabled(False)
        try:
            y_g = ref_blk(xg)
        finally:
            fused_block.set_enabled(True)
    assert rel(y_f, y_g) < 2e-5, rel(y_f, y_g)
    assert rel(y_f, y_t) < 1e-4, rel(y_f, y_t)
    cs = ops.act_cs(y_f)
    full = torch.as_strided(y_f, (n, cs, h, w), y_f.stride())
    assert cs == c or float(full[:, c:].abs().max()) == 0.0
    if norm == 'batch':      # running statistics and the batch counter of every norm module advanced exactly like torch's
        tsd = twin.state_dict()
        for k, v in blk.state_dict().items():
            if 'running_' in k:
                assert rel(v, tsd[k]) < 1e-5, k
            if 'num_batches_tracked' in k:
                assert int(v) == int(tsd[k]) == 1, k


@pytest.mark.parametrize('norm,padding,shape', [('batch', 'reflect', (4, 77, 48, 64)), ('instance', 'reflect', (4, 77, 48, 64)),
                                                ('batch', 'zero', (3, 77, 50, 70)), ('batch', 'reflect', (8, 40, 32, 48))])
def test_fused_block_backward_matches_torch(norm, padding, shape):
    """Input gradient and every parameter gradient of the fused path against stock PyTorch autograd on the host (fp32)."""
    from cat_amd import _lib, fused_block, ops
    _lib.load()
    dev = torch.device('cuda:0')
    n, c, h, w = shape
    res, dw = ((11, 12, 18), (15, 15, 12)) if c == 77 else ((7, 0, 9), (16, 5, 0))
    blk = _block(norm, dev, c, res, dw, padding)
    x = detfill.normal((n, c, h, w), 5)
    gy = detfill.normal((n, c, h, w), 6)
    twin, twin_fwd = _torch_twin(blk)
    xr = x.clone().requires_grad_(True)
    twin_fwd(xr).backward(gy)
    xg = ops.to_nhwc(x.to(dev)).detach().requires_grad_(True)
    assert fused_block.applicable(blk, xg)
    y = blk(xg)
    y.backward(ops.to_nhwc(gy.to(dev)))
    torch.cuda.synchronize()
    assert rel(xg.grad, xr.grad) < 2e-4, rel(xg.grad, xr.grad)
    tgrads = dict(twin.named_parameters())
    # gradients that are zero in exact arithmetic (a conv bias or a 1x1 depthwise scale directly in front of an InstanceNorm) are pure
    # round-off in both implementations: every tensor is judged against max(its own scale, 1e-3 of the largest gradient)
    top = max(float(q.grad.abs().max()) for q in tgrads.values())
    bad = {}
    for k, q in blk.named_parameters():
        ref = tgrads[k].grad
        assert q.grad is not None, k
        if norm == 'instance' and (k.endswith('.bias') or k == 'dw_ops.0.2.0.weight'):
            continue        # per-channel shift / scale directly in front of an InstanceNorm: zero gradient in exact arithmetic
        err = float((q.grad.detach().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * top)
        if err > 5e-4:
            bad[k] = err
    assert not bad, bad


def test_fused_block_backward_into_optimizer_buffers_with_one_channel_branches():
    """As in the training step the gradients are slices of FusedAdam's flat buffer, where a conv weight with ONE input channel is stored
    unpadded ([O][1][1][1]); depthwise branches pruned to a single channel are scattered there from the K-concatenated weight gradient
    without touching the neighbouring parameters (every gradient against stock PyTorch autograd on the host)."""
    from cat_amd import _lib, fused_block, ops
    from cat_amd.optim import FusedAdam
    _lib.load()
    dev = torch.device('cuda:0')
    n, c, h, w = 8, 40, 32, 48
    blk = _block('batch', dev, c, (1, 0, 9), (1, 5, 1), 'reflect')
    twin, twin_fwd = _torch_twin(blk)
    FusedAdam(list(blk.parameters()), lr=0.0).zero_grad()
    x, gy = detfill.normal((n, c, h, w), 5), detfill.normal((n, c, h, w), 6)
    xr = x.clone().requires_grad_(True)
    twin_fwd(xr).backward(gy)
    xg = ops.to_nhwc(x.to(dev)).detach().requires_grad_(True)
    assert fused_block.applicable(blk, xg)
    blk(xg).backward(ops.to_nhwc(gy.to(dev)))
    torch.cuda.synchronize()
    assert rel(xg.grad, xr.grad) < 2e-4, rel(xg.grad, xr.grad)
    tgrads = dict(twin.named_parameters())
    top = max(float(q.grad.abs().max()) for q in tgrads.values())
    bad = {}
    for k, q in blk.named_parameters():
        ref = tgrads[k].grad
        assert q.grad is not None and q.grad.data_ptr() == q._cat_grad_view.data_ptr(), k
        err = float((q.grad.detach().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * top)
        if err > 5e-4:
            bad[k] = err
    assert not bad, bad


def test_fused_block_plan_follows_replaced_parameters():
    """A parameter object replaced by a new one of the same shape (weight transfer, `m.weight = nn.Parameter(...)`) must receive its
    gradient: the cached plan keys gradient targets by parameter identity and is rebuilt when the identities change."""
    from cat_amd import _lib, fused_block, ops
    _lib.load()
    dev = torch.device('cuda:0')
    blk = _block('batch', dev)
    x = ops.to_nhwc(detfill.normal((4, 77, 48, 64), 5).to(dev))
    gy = ops.to_nhwc(detfill.normal((4, 77, 48, 64), 6).to(dev))
    xg = x.detach().requires_grad_(True)
    blk(xg).backward(gy)
    old = blk.res_ops[0][1][1].weight
    g_old = old.grad.clone()
    blk.res_ops[0][1][1].weight = torch.nn.Parameter(old.detach().clone())
    blk.pw_bn.bias = torch.nn.Parameter(blk.pw_bn.bias.detach().clone())
    blk.zero_grad()
    xg = x.detach().requires_grad_(True)
    assert fused_block.applicable(blk, xg)
    blk(xg).backward(gy)
    torch.cuda.synchronize()
    new = blk.res_ops[0][1][1].weight
    assert new.grad is not None and blk.pw_bn.bias.grad is not None
    assert rel(new.grad, g_old) < 1e-4      # (running statistics moved between the two steps: batch statistics did not)
