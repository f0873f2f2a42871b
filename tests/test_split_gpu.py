"""GPU: the opt-in split-bf16 backward tiles (csrc/conv_split.hip, switch CAT_MFMA=bf16x3 / ops.set_mfma_split) against the exact-fp32 path and
against an fp64 evaluation on the host, on the shapes of PatchGAN's wide 4x4 layers (reference models/modules/discriminators.py:38-76) and ragged
ones.  The split form drops <= 2^-16 of a product: its results are held to 2e-5 of sum |terms| (the fp32 path sits at ~1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detfill

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda', 0)


@pytest.fixture()
def split():
    from cat_amd import ops
    old = ops.set_mfma_split(True)
    yield ops
    ops.set_mfma_split(old)


def test_split_planes_are_the_bf16_parts(split):
    ops = split
    x = (detfill.normal((3, 128, 9, 7), 11) * torch.logspace(-3, 3, 3 * 128 * 9 * 7).reshape(3, 128, 9, 7)).to(dev())
    xn = ops.to_nhwc(x)
    planes = ops.split_planes(xn)
    flat = torch.as_strided(xn, (xn.numel(),), (1,), xn.storage_offset())
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.float()).to(torch.bfloat16)
    assert torch.equal(planes[:xn.numel()].view(torch.int16), hi.view(torch.int16))
    assert torch.equal(planes[xn.numel():].view(torch.int16), lo.view(torch.int16))


CASES = [   # cin, cout, k, stride, pad, N, H, W
    (128, 256, 4, 2, 1, 2, 32, 32),      # PatchGAN layer 2
    (256, 512, 4, 2, 1, 2, 16, 16),      # layer 3
    (512, 512, 4, 1, 1, 2, 16, 16),      # layer 4 (stride 1: 15 x 15 output)
    (128, 128, 4, 2, 1, 3, 20, 36),      # ragged lattice, partial tiles
    (256, 128, 4, 1, 1, 1, 11, 23),      # stride 1, odd sizes
]


@pytest.mark.parametrize('case', CASES)
def test_backward_tiles_in_split_form(split, case):
    ops = split
    from cat_amd import _lib as L
    cin, cout, k, s, p, n, h, w = case
    x = detfill.normal((n, cin, h, w), 21)
    wt = detfill.normal((cout, cin, k, k), 22) * (cin * k * k) ** -0.5
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    gy = detfill.normal((n, cout, ho, wo), 23)

    def run(on):
        ops.set_mfma_split(on)
        ops.STATS['split_dgrad'] = ops.STATS['split_wgrad'] = 0
        xd = ops.to_nhwc(x.to(dev())).requires_grad_(True)
        wd = ops.padded_weight_like((cout, cin, k, k), dev())
        wd.copy_(wt.to(dev()))
        wd.requires_grad_(True)
        y = ops.Conv2dFn.apply(xd, wd, None, s, p, L.PAD_ZERO, L.ACT_NONE, 0.0)
        y.backward(ops.to_nhwc(gy.to(dev())))
        return xd.grad.detach().cpu().double(), wd.grad.detach().cpu().double(), dict(ops.STATS)

    dx1, dw1, st1 = run(True)
    dx0, dw0, st0 = run(False)
    assert st1['split_dgrad'] == 1 and st1['split_wgrad'] == 1 and st0['split_dgrad'] == 0 and st0['split_wgrad'] == 0
    # fp64 on the host, and the magnitude each sum is made of
    x64, w64, g64 = x.double().requires_grad_(True), wt.double().requires_grad_(True), gy.double()
    F.conv2d(x64, w64, None, s, p).backward(g64)
    xa, wa = x.double().abs().requires_grad_(True), wt.double().abs().requires_grad_(True)
    F.conv2d(xa, wa, None, s, p).backward(g64.abs())
    for name, got1, got0, ref, mag in (('dx', dx1, dx0, x64.grad, xa.grad), ('dw', dw1, dw0, w64.grad, wa.grad)):
        e1 = float(((got1 - ref).abs() / mag.clamp_min(1e-30)).max())
        e0 = float(((got0 - ref).abs() / mag.clamp_min(1e-30)).max())
        print('%s %s: split %.2e   fp32 %.2e   of sum |terms|' % (case, name, e1, e0))
        assert e1 < 2e-5 and e0 < 5e-6, (name, e1, e0)


def test_discriminator_step_under_the_switch(split):
    """NLayerDiscriminator ndf 128 forward + backward (BatchNorm, LeakyReLU): every parameter gradient and the input gradient with the switch on
    against the switch off -- the layers 2..4 take the split tiles, the rest is untouched."""
    ops = split
    from cat_amd import networks, synthetic
    opt = synthetic.default_options(norm='batch', track=True, ndf=128, gpu_ids=[0])
    torch.manual_seed(5)
    D = networks.define_D(6, 128, 'n_layers', 3, 'batch', 'normal', 0.02, [0], opt=opt).train()
    sd = {k: v.clone() for k, v in D.state_dict().items()}
    x = detfill.images((2, 6, 128, 128), 31)

    def run(on):
        ops.set_mfma_split(on)
        ops.STATS['split_dgrad'] = ops.STATS['split_wgrad'] = 0
        D.load_state_dict(sd)
        D.zero_grad(set_to_none=True)
        xd = ops.to_nhwc(x.to(dev())).requires_grad_(True)
        out = D(xd)
        out.backward(ops.to_nhwc(detfill.normal(tuple(out.shape), 32).to(dev())))
        return {k: p.grad.detach().cpu().double() for k, p in D.named_parameters()}, xd.grad.detach().cpu().double(), dict(ops.STATS)

    g1, dx1, st1 = run(True)
    g0, dx0, st0 = run(False)
    assert st1['split_dgrad'] == 3 and st1['split_wgrad'] == 3 and st0['split_dgrad'] == 0
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    worst = max(rel(g1[k], g0[k]) for k in g0)
    print('discriminator gradients, switch on vs off: worst tensor %.2e, input gradient %.2e' % (worst, rel(dx1, dx0)))
    assert worst < 1e-3 and rel(dx1, dx0) < 1e-3
