"""GPU parity of every HIP kernel family against the CPU oracle ops (stock fp32 ATen on the host), called through the
C-ABI (cat_amd.ops -> ctypes -> libcat_hip.so).  Tolerance: 1e-3 relative (north_star); observed ~1e-6..1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detfill, ref_cpu

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope='module')
def dev():
    from cat_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _nhwc(x, dev, requires_grad=False):
    from cat_amd import ops
    y = ops.to_nhwc(x.to(dev))
    if requires_grad:
        y = y.detach().requires_grad_(True)
    return y


def _cl(w, dev):
    w = w.to(dev)
    if w.dim() == 4 and w.shape[1] > 1:
        w = w.contiguous(memory_format=torch.channels_last)
    return w.detach().requires_grad_(True)


CONV_CASES = [
    # cin, cout, k, stride, pad, reflect, act, N, H, W
    (3, 16, 7, 1, 3, True, 0, 2, 20, 24),
    (16, 35, 3, 2, 1, False, 0, 2, 20, 24),
    (54, 7, 5, 1, 2, True, 0, 2, 16, 16),
    (54, 13, 1, 1, 0, False, 0, 3, 9, 11),
    (13, 54, 5, 1, 2, True, 0, 2, 16, 12),
    (6, 128, 4, 2, 1, False, 2, 2, 32, 32),
    (128, 256, 4, 2, 1, False, 0, 2, 16, 16),
    (64, 1, 4, 1, 1, False, 0, 2, 9, 9),
    (16, 3, 7, 1, 3, True, 3, 1, 18, 18),
    (256, 42, 3, 1, 1, True, 0, 1, 12, 12),
    (42, 256, 3, 1, 1, True, 0, 1, 12, 12),
    (82, 100, 3, 1, 1, False, 0, 1, 10, 10),
    (130, 17, 1, 1, 0, False, 0, 1, 8, 8),
    (5, 5, 3, 2, 1, False, 0, 1, 7, 9),       # odd sizes, stride 2
    # few pixels, deep reduction (SPADE generator heads): split-K forward and dgrad
    (1024, 170, 5, 1, 2, False, 2, 4, 4, 8),
    (170, 1024, 3, 1, 1, False, 0, 2, 8, 16),
    (256, 23, 5, 1, 2, False, 0, 4, 4, 8),
    (512, 31, 1, 1, 0, False, 0, 2, 8, 8),
    (7, 512, 5, 1, 2, False, 0, 4, 4, 8),
    (96, 16, 3, 1, 1, True, 0, 1, 6, 6),
    # wide layers: 32-deep dgrad / wgrad chunks (tails in K, Cin, Cout; reflect; direct single-slice wgrad), channel-split head
    (128, 64, 3, 1, 1, True, 0, 2, 9, 7),
    (132, 48, 3, 1, 1, False, 0, 1, 10, 12),
    (100, 132, 3, 1, 1, True, 0, 2, 9, 9),
    (160, 112, 1, 1, 0, False, 0, 1, 5, 5),
    (256, 128, 4, 1, 1, False, 2, 3, 12, 10),
    (1024, 1, 4, 1, 1, False, 0, 4, 12, 12),
    # direct-to-LDS forward / dgrad / wgrad tiles: Cin % 128 == 0, output rows of 64 and 40 -> 2 row segments, ragged last segment excluded;
    # stride 1 with a 31-wide plane (the PatchGAN's 512 -> 1024 layer in small)
    (128, 160, 4, 2, 1, False, 2, 1, 12, 128),
    (256, 128, 4, 1, 1, False, 0, 2, 11, 32),
    (128, 130, 3, 2, 1, False, 0, 2, 9, 64),
    (512, 1, 4, 1, 2, False, 0, 2, 9, 17),
    # ... and rows that END in a partial segment (63 and 50 output columns: the 512 x 512 PatchGAN's stride-1 layers)
    (128, 128, 4, 1, 1, False, 0, 1, 8, 64),
    (128, 160, 4, 2, 1, False, 2, 1, 10, 100),
    # PatchGAN's first layer: the image gradient through the class-per-wave 4x4 / stride 2 dgrad (3 and 6 channels, ragged and odd planes)
    (3, 64, 4, 2, 1, False, 2, 2, 70, 50),
    (6, 20, 4, 2, 1, False, 0, 1, 33, 37),
    (6, 128, 4, 2, 1, False, 2, 2, 64, 96),
    # 1 x 1 layers with few channels over many pixels: the pixel-streaming weight gradient (csrc/conv_pwgrad.hip) -- two 64-channel groups
    # of x, a pixel count that is no multiple of the 32-pixel step, 65 .. 128 output channels (two accumulator groups), three x groups
    (77, 60, 1, 1, 0, False, 0, 4, 64, 64),
    (12, 16, 1, 1, 0, False, 0, 5, 50, 70),
    (36, 100, 1, 1, 0, False, 0, 2, 96, 96),
    (130, 7, 1, 1, 0, False, 0, 1, 128, 130),
    (4, 64, 1, 1, 0, False, 0, 3, 61, 67),
]


@pytest.mark.parametrize('cin,cout,k,stride,pad,reflect,act,n,h,w', CONV_CASES)
def test_conv2d_fwd_bwd(dev, cin, cout, k, stride, pad, reflect, act, n, h, w):
    from cat_amd import ops
    x = detfill.normal((n, cin, h, w), 1)
    wt = detfill.normal((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
    b = detfill.normal((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xp = F.pad(xr, (pad,) * 4, mode='reflect') if reflect and pad else xr
    yr = F.conv2d(xp, wr, br, stride=stride, padding=0 if reflect else pad)
    if act == 2:
        yr = F.leaky_relu(yr, 0.2)
    elif act == 3:
        yr = torch.tanh(yr)
    gy = detfill.normal(tuple(yr.shape), 4)
    yr.backward(gy)

    xg, wg, bg = _nhwc(x, dev, True), _cl(wt, dev), b.to(dev).requires_grad_(True)
    y = ops.Conv2dFn.apply(xg, wg, bg, stride, pad, 1 if reflect else 0, act, 0.2)
    assert y.shape == yr.shape
    assert rel(y, yr) < TOL
    # padding channels of the NHWC buffer stay zero
    cs = ops.act_cs(y)
    full = torch.as_strided(y, (y.shape[0], cs, y.shape[2], y.shape[3]), y.stride())
    assert float(full[:, cout:].abs().max()) == 0.0 if cs > cout else True
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wg.grad, wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


def test_pointwise_wgrad_takes_the_streaming_kernel(dev):
    """The 1 x 1 cases above must run cat_pw::pwgrad_kernel (family conv_pwgrad), not fall through to the general kernel."""
    from cat_amd import _lib, ops
    lib = _lib.load()
    lib.cat_prof_enable(1)
    try:
        x = _nhwc(detfill.normal((4, 12, 64, 64), 1), dev, True)
        wg = _cl(detfill.normal((16, 12, 1, 1), 2), dev)
        y = ops.Conv2dFn.apply(x, wg, None, 1, 0, 0, 0, 0.2)
        y.backward(_nhwc(detfill.normal((4, 16, 64, 64), 3), dev))
        torch.cuda.synchronize()
        fam = _families()
    finally:
        lib.cat_prof_enable(0)
    assert fam.get('conv_pwgrad', 0) == 1, fam


def _families():
    import ctypes as C
    from cat_amd import _lib
    lib = _lib.load()
    n = lib.cat_prof_collect()
    name, cnt, ms, fl = C.create_string_buffer(64), C.c_int64(), C.c_double(), C.c_double()
    out = {}
    for i in range(n):
        lib.cat_prof_family(i, name, 64, C.byref(cnt), C.byref(ms), C.byref(fl))
        out[name.value.decode()] = cnt.value
    return out


# stride-1 3x3 / 5x5 layers on planes large enough for the LDS-tile kernels with packed filters (csrc/conv_pk.hip): the pruned
# student's ragged widths, the teacher's 256 <-> 42 pairs, N blocks (Cout > 128), ragged plane edges, zero and reflect padding
TCONV_CASES = [
    # cin, cout, k, reflect, act, N, H, W
    (77, 18, 5, True, 0, 4, 48, 64), (18, 77, 5, True, 0, 4, 48, 64), (77, 12, 3, False, 2, 4, 48, 64), (9, 77, 5, False, 0, 3, 50, 70),
    (256, 42, 5, True, 0, 4, 56, 64), (42, 256, 3, True, 0, 4, 56, 64), (23, 130, 3, False, 0, 3, 61, 67), (16, 16, 3, True, 3, 4, 48, 64),
    # the LDS-tile weight gradient (csrc/conv_twgrad.hip): wide x / narrow dy and the reverse, 5x5 and 3x3, one and two narrow tiles,
    # reflect and zero padding, ragged plane edges, a wide side of three tiles
    (77, 15, 5, True, 0, 4, 64, 64), (15, 77, 5, True, 0, 4, 64, 64), (77, 9, 5, False, 0, 3, 61, 67), (13, 77, 3, False, 0, 3, 61, 67),
    (77, 18, 3, True, 0, 4, 64, 64), (18, 77, 3, True, 0, 4, 64, 64), (40, 12, 3, True, 0, 4, 64, 48),
]


@pytest.mark.parametrize('cin,cout,k,reflect,act,n,h,w', TCONV_CASES)
def test_conv2d_lds_tile_fwd_bwd(dev, cin, cout, k, reflect, act, n, h, w):
    from cat_amd import _lib, ops
    pad = (k - 1) // 2
    assert ops.tconv_applicable(n, h, w, cout, k, k, 1, pad)
    x = detfill.normal((n, cin, h, w), 1)
    wt = detfill.normal((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
    b = detfill.normal((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xp = F.pad(xr, (pad,) * 4, mode='reflect') if reflect else xr
    yr = F.conv2d(xp, wr, br, padding=0 if reflect else pad)
    yr = {0: yr, 2: F.leaky_relu(yr, 0.2), 3: torch.tanh(yr)}[act]
    gy = detfill.normal(tuple(yr.shape), 4)
    yr.backward(gy)
    xg, bg = _nhwc(x, dev, True), b.to(dev).requires_grad_(True)
    wg = ops.padded_weight_like(wt.shape, dev)
    wg.copy_(wt)
    wg.requires_grad_(True)
    lib = _lib.load()
    lib.cat_prof_enable(1)
    y = ops.Conv2dFn.apply(xg, wg, bg, 1, pad, 1 if reflect else 0, act, 0.2)
    y.backward(_nhwc(gy, dev))
    torch.cuda.synchronize()
    fam = _families()
    lib.cat_prof_enable(0)
    assert fam.get('conv_tconv', 0) == 2, fam          # forward + input gradient both ran on the LDS-tile kernel
    tl = lambda c: ((c + 3) // 4 * 4 + 15) // 16
    narrow, wide = sorted((tl(cin), tl(cout)))
    if max(cin, cout) <= 80 and narrow <= (1 if k == 5 else 2) and wide >= 3 and n * ((h + 7) // 8) * ((w + 7) // 8) >= 128:
        assert fam.get('conv_twgrad', 0) == 1, fam     # ... and the weight gradient on the LDS-tile wgrad kernel
    assert rel(y, yr) < TOL
    cs = ops.act_cs(y)
    full = torch.as_strided(y, (y.shape[0], cs, y.shape[2], y.shape[3]), y.stride())
    assert cs == cout or float(full[:, cout:].abs().max()) == 0.0
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wg.grad, wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL
    # a changed weight must be re-packed (the cache keys on torch's version counter / the optimizer epoch)
    with torch.no_grad():
        wg.mul_(2.0)
        y2 = ops.Conv2dFn.apply(xg.detach(), wg, None, 1, pad, 1 if reflect else 0, 0, 0.0)
        y1 = ops.Conv2dFn.apply(xg.detach(), wg * 0.5, None, 1, pad, 1 if reflect else 0, 0, 0.0)
    assert rel(y2, 2.0 * y1) < 1e-6


@pytest.mark.parametrize('reflect', [True, False])
def test_tconv_multi_source_branch_sum(dev, reflect):
    """K-concatenated sum of six convs (k = 1, 3, 5, 1, 1, 1) over channel slices of one hidden buffer, with the normalise + ReLU of
    the train-mode norm applied while staging: InvertedResidualChannels' branch sum (inception_modules.py:230-236) as one launch."""
    from cat_amd import _lib as L, ops, tconv
    n, h, w, cout = 2, 24, 40, 77
    ms, kss = [11, 12, 18, 15, 15, 12], [1, 3, 5, 1, 1, 1]
    offs = np.cumsum([0] + [tconv.cs4(m) for m in ms])
    hc = int(offs[-1])
    hbuf, sc_all, sh_all = torch.zeros(n, h, w, hc), torch.zeros(hc), torch.zeros(hc)
    ref = torch.zeros(n, cout, h, w)
    packs = []
    for bi, (m, k) in enumerate(zip(ms, kss)):
        hb = detfill.normal((n, m, h, w), 20 + bi)
        sc, sh = detfill.normal((m,), 40 + bi).abs() + 0.5, detfill.normal((m,), 60 + bi, 0.3)
        wt = detfill.normal((cout, m, k, k), 80 + bi, 1.0 / np.sqrt(m * k * k * 6))
        a = F.relu(hb * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        p = (k - 1) // 2
        ref += F.conv2d(F.pad(a, (p,) * 4, mode='reflect') if (reflect and p) else a, wt, None, padding=0 if reflect else p)
        o = int(offs[bi])
        hbuf[..., o:o + m] = hb.permute(0, 2, 3, 1)
        sc_all[o:o + m], sh_all[o:o + m] = sc, sh
        wg = ops.padded_weight_like(wt.shape, dev)
        wg.copy_(wt)
        packs.append(tconv.pack(wg, tconv.FWD))
    hg, scg, shg = hbuf.to(dev), sc_all.to(dev), sh_all.to(dev)
    segs, poff = [], 0
    for bi, (m, k) in enumerate(zip(ms, kss)):
        o = int(offs[bi])
        segs.append(tconv.Segment(hg, k, (k - 1) // 2, reflect, poff, c4=tconv.cs4(m), scale=scg[o:], shift=shg[o:], act=L.ACT_RELU, xcs=hc,
                                  ptr=hg.data_ptr() + 4 * o))
        poff += packs[bi].numel()
    b = detfill.normal((cout,), 99, 0.1)
    y = ops.empty_act(n, cout, h, w, dev)
    tconv.run(segs, torch.cat(packs), b.to(dev), y, cout, n, h, w, h, w)
    assert rel(y, ref + b.view(1, -1, 1, 1)) < TOL


@pytest.mark.parametrize('cin,cout,n,h,w', [(54, 33, 2, 8, 8), (256, 128, 1, 8, 6), (33, 16, 2, 9, 7), (4, 4, 1, 3, 3)])
def test_conv_transpose2d(dev, cin, cout, n, h, w):
    from cat_amd import ops
    x = detfill.normal((n, cin, h, w), 5)
    wt = detfill.normal((cin, cout, 3, 3), 6, 1.0 / np.sqrt(cin * 9))
    b = detfill.normal((cout,), 7, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2, padding=1, output_padding=1)
    gy = detfill.normal(tuple(yr.shape), 8)
    yr.backward(gy)
    xg, wg, bg = _nhwc(x, dev, True), _cl(wt, dev), b.to(dev).requires_grad_(True)
    y = ops.ConvTranspose2dFn.apply(xg, wg, bg, 2, 1, 1)
    assert y.shape == yr.shape and rel(y, yr) < TOL
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wg.grad, wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


@pytest.mark.parametrize('c,k,reflect,n,h,w', [(11, 3, True, 2, 12, 10), (14, 5, True, 2, 9, 9), (8, 1, False, 2, 6, 6), (42, 5, True, 1, 16, 16),
                                               (17, 3, False, 1, 7, 5), (170, 5, False, 2, 8, 16), (128, 7, False, 1, 9, 9)])
def test_depthwise_conv(dev, c, k, reflect, n, h, w):
    from cat_amd import ops
    pad = (k - 1) // 2
    x = detfill.normal((n, c, h, w), 9)
    wt = detfill.normal((c, 1, k, k), 10, 1.0 / k)
    b = detfill.normal((c,), 11, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xp = F.pad(xr, (pad,) * 4, mode='reflect') if reflect and pad else xr
    yr = F.conv2d(xp, wr, br, padding=0 if reflect else pad, groups=c)
    gy = detfill.normal(tuple(yr.shape), 12)
    yr.backward(gy)
    xg, wg, bg = _nhwc(x, dev, True), wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.DwConv2dFn.apply(xg, wg, bg, pad, 1 if reflect else 0)
    assert rel(y, yr) < TOL
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wg.grad, wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


@pytest.mark.parametrize('mode,c,n,h,w,act', [('instance', 54, 2, 16, 16, 1), ('instance', 7, 3, 9, 5, 1), ('batch', 77, 4, 8, 8, 1),
                                              ('batch', 256, 2, 6, 6, 2), ('instance', 1024, 1, 5, 5, 2), ('batch', 16, 2, 64, 64, 0),
                                              ('instance', 1100, 1, 4, 4, 0), ('batch', 42, 3, 10, 12, 4), ('instance', 19, 2, 9, 9, 4)])
def test_norm_fwd_bwd(dev, mode, c, n, h, w, act):
    from cat_amd import ops, _lib as L
    x = detfill.normal((n, c, h, w), 13) * 2.0 + 3.0      # non-zero mean: exercises the shifted-sum variance
    ga = (3.0 if act == 4 else 1.0) * (1.0 + 0.2 * detfill.normal((c,), 14))
    be = 0.1 * detfill.normal((c,), 15)
    xr, gr, br = x.clone().requires_grad_(True), ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    if mode == 'instance':
        yr = F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    else:
        yr = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    if act == 4:      # nn.ReLU6 (get_active_fn('nn.ReLU6'), reference inception_modules.py:12-19): scale so that both bounds are hit
        yr = F.relu6(yr)
    else:
        yr = F.relu(yr) if act == 1 else (F.leaky_relu(yr, 0.2) if act == 2 else yr)
    gy = detfill.normal(tuple(yr.shape), 16)
    yr.backward(gy)
    xg = _nhwc(x, dev, True)
    gg, bg = ga.to(dev).requires_grad_(True), be.to(dev).requires_grad_(True)
    rmg, rvg = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    y = ops.NormActFn.apply(xg, gg, bg, rmg if mode == 'batch' else None, rvg if mode == 'batch' else None,
                            L.NORM_INSTANCE if mode == 'instance' else L.NORM_BATCH, 1e-5, 0.1, act, 0.2)
    assert rel(y, yr) < TOL
    if mode == 'batch':
        assert rel(rmg, rm) < TOL and rel(rvg, rv) < TOL
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < 5 * TOL
    assert rel(gg.grad, gr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


@pytest.mark.parametrize('reflect', [True, False])
def test_dwconv_multi_fwd(dev, reflect):
    """cat_dwconv2d_multi_fwd: depthwise convs of kernel sizes 1 / 3 / 5 (+ a copied slice) over adjacent channel slices of one buffer as ONE
    launch, against F.conv2d(groups=C) per slice on the host -- the frozen teacher's block in small (42-channel slices in 44-channel slots)."""
    import ctypes as C
    from cat_amd import ops, _lib as L
    n, h, w, m = 2, 11, 9, 42
    sz = 44
    hc = 4 * sz
    x = F.relu(detfill.normal((n, hc, h, w), 71))
    for s_ in range(4):
        x[:, s_ * sz + m:(s_ + 1) * sz] = 0.0          # padding channels of a slot hold zeros
    frame = torch.zeros(25, hc)
    bias = torch.zeros(hc)
    want = torch.zeros(n, hc, h, w)
    want[:, :m] = x[:, :m]                             # slot 0: the copied slice (k = 1, centre weight 1)
    frame[12, :m] = 1.0
    ks = [1] * (hc // 4)
    for s_, k in ((1, 1), (2, 3), (3, 5)):
        wt = detfill.normal((m, 1, k, k), 80 + k) / k
        b = 0.1 * detfill.normal((m,), 90 + k)
        xs = x[:, s_ * sz:s_ * sz + m]
        p = k // 2
        xp = F.pad(xs, (p, p, p, p), mode='reflect') if (reflect and p) else xs
        want[:, s_ * sz:s_ * sz + m] = F.relu(F.conv2d(xp, wt, b, 1, 0 if (reflect and p) else p, 1, m))
        o2 = 2 - p
        for ky in range(k):
            for kx in range(k):
                frame[(o2 + ky) * 5 + o2 + kx, s_ * sz:s_ * sz + m] = wt[:, 0, ky, kx]
        bias[s_ * sz:s_ * sz + m] = b
        for q in range(s_ * sz // 4, (s_ + 1) * sz // 4):
            ks[q] = k
    xg = _nhwc(x, dev)
    y = ops.empty_act(n, hc, h, w, dev)
    g = L.DwMulti()
    g.N, g.H, g.W, g.nq, g.xcs, g.ycs, g.reflect, g.act, g.slope = n, h, w, hc // 4, hc, hc, int(reflect), L.ACT_RELU, 0.0
    for q, k in enumerate(ks):
        g.ks[q] = k
    frame_d, bias_d = frame.to(dev).contiguous(), bias.to(dev)      # held until the launch has run (a temporary's block would be re-used)
    L.call('cat_dwconv2d_multi_fwd', C.byref(g), ops._p(xg), ops._p(frame_d), ops._p(bias_d), ops._p(y), ops._stream())
    torch.cuda.synchronize()
    assert rel(y, want) < TOL
    assert float(y.cpu()[:, m:sz].abs().max()) == 0.0      # padding channels stay exact zeros


@pytest.mark.parametrize('affine', [True, False])
def test_instance_norm_with_running_stats(dev, affine):
    """nn.InstanceNorm2d(track_running_stats=True) (reference models/networks.py:29-64 with --norm instance --norm_track_running_stats):
    instance statistics + batch-averaged running statistics (unbiased variance) in training, the running statistics in evaluation --
    the module against torch's own on the host, two training steps then an eval forward."""
    from cat_amd import nn as cnn
    c, n, h, w = 19, 3, 9, 7
    ref = torch.nn.InstanceNorm2d(c, affine=affine, track_running_stats=True)
    mod = cnn.InstanceNorm2d(c, affine=affine, track_running_stats=True)
    if affine:
        with torch.no_grad():
            ref.weight.copy_(1.0 + 0.2 * detfill.normal((c,), 31))
            ref.bias.copy_(0.1 * detfill.normal((c,), 32))
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(dev)
    ref.train(), mod.train()
    for it in range(2):
        x = detfill.normal((n, c, h, w), 40 + it) * (1.5 + it) + 0.7
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        gy = detfill.normal(tuple(yr.shape), 50 + it)
        yr.backward(gy)
        xg = _nhwc(x, dev, True)
        y = mod(xg)
        assert rel(y, yr) < TOL
        y.backward(_nhwc(gy, dev))
        assert rel(xg.grad, xr.grad) < 5 * TOL
        assert rel(mod.running_mean, ref.running_mean) < TOL and rel(mod.running_var, ref.running_var) < TOL
        assert int(mod.num_batches_tracked) == int(ref.num_batches_tracked)      # torch's _InstanceNorm never counts
    ref.eval(), mod.eval()
    x = detfill.normal((n, c, h, w), 60)
    with torch.no_grad():
        assert rel(mod(_nhwc(x, dev)), ref(x)) < TOL


def test_bn_eval_affine(dev):
    from cat_amd import ops
    c = 42
    x = detfill.normal((2, c, 8, 8), 17)
    ga, be = 1.0 + 0.2 * detfill.normal((c,), 18), 0.1 * detfill.normal((c,), 19)
    rm, rv = 0.1 * detfill.normal((c,), 20), 0.5 + torch.rand(c)
    yr = F.relu(F.batch_norm(x, rm, rv, ga, be, False, 0.1, 1e-5))
    scale, shift = ops.bn_fold(ga.to(dev), be.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    with torch.no_grad():
        y = ops.affine_act(_nhwc(x, dev), scale, shift, 1, 0.0)
    assert rel(y, yr) < TOL


@pytest.mark.parametrize('n,cx,cy,h,w', [(2, 7, 11, 6, 5), (3, 54, 256, 16, 16), (16, 56, 256, 16, 16), (17, 5, 9, 8, 8), (1, 8, 8, 4, 4)])
def test_ka(dev, n, cx, cy, h, w):
    from cat_amd import ops
    X, Y = detfill.normal((n, cx, h, w), 100 + n), detfill.normal((n, cy, h, w), 200 + n)
    Xr = X.clone().requires_grad_(True)
    vr = ref_cpu.ka(Xr, Y)
    (-1.3 * vr).backward()
    Xg = _nhwc(X, dev, True)
    v = ops.ka(Xg, _nhwc(Y, dev))
    assert abs(v.item() - vr.item()) < 1e-5
    torch.autograd.backward([v], [torch.full((), -1.3, device=dev)])
    if n == 1:      # KA == 1 identically: the gradient is pure round-off on both sides
        assert float(Xg.grad.abs().max()) < 1e-6 and float(Xr.grad.abs().max()) < 1e-6
        return
    assert rel(Xg.grad, Xr.grad) < 5 * TOL


def test_ka_matches_reference_golden(dev):
    import helpers as H
    from cat_amd import ops
    g = H.load('small_ops.npz')
    for n in (2, 3, 16):
        X = detfill.normal((n, 7, 6, 5), 100 + n)
        Y = detfill.normal((n, 11, 6, 5), 200 + n)
        Xg = _nhwc(X, dev, True)
        v = ops.ka(Xg, _nhwc(Y, dev))
        v.backward()
        assert abs(v.item() - float(g[f'ka{n}'])) < 1e-5
        assert rel(Xg.grad, torch.from_numpy(g[f'ka{n}_grad'])) < 5 * TOL


def test_gan_and_recon_losses(dev):
    import helpers as H
    from cat_amd.loss import GANLoss, L1Loss, MSELoss
    g = H.load('small_ops.npz')
    pred = detfill.normal((2, 1, 6, 6), 300, 1.5)
    for mode in ('hinge', 'lsgan', 'vanilla', 'wgangp'):
        crit = GANLoss(mode)
        for real in (True, False):
            p = _nhwc(pred, dev, True)
            l = crit(p, real, for_discriminator=True)
            l.backward()
            assert abs(l.item() - float(g[f'gan_{mode}_D_{int(real)}'])) < 1e-6
            assert rel(p.grad, torch.from_numpy(g[f'gan_{mode}_D_{int(real)}_grad'])) < 1e-6
        p = _nhwc(pred, dev, True)
        l = crit(p, True, for_discriminator=False)
        l.backward()
        assert abs(l.item() - float(g[f'gan_{mode}_G'])) < 1e-6
        assert rel(p.grad, torch.from_numpy(g[f'gan_{mode}_G_grad'])) < 1e-6
    a, b = detfill.normal((2, 3, 16, 16), 301), detfill.normal((2, 3, 16, 16), 302)
    for crit, ref in ((L1Loss(), F.l1_loss), (MSELoss(), F.mse_loss)):
        ar = a.clone().requires_grad_(True)
        lr = ref(ar, b)
        lr.backward()
        ag = _nhwc(a, dev, True)
        l = crit(ag, _nhwc(b, dev))
        l.backward()
        assert abs(l.item() - lr.item()) < 1e-6 and rel(ag.grad, ar.grad) < 1e-6


def test_adam_matches_torch(dev):
    from cat_amd.optim import FusedAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(detfill.normal(s, 400 + i)) for i, s in enumerate([(7, 5, 3, 3), (11,), (4, 1, 5, 5)])]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    oref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    gp = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ps]
    opt = FusedAdam(gp, lr=2e-4, betas=(0.5, 0.999))
    for step in range(3):
        grads = [detfill.normal(tuple(p.shape), 500 + 10 * step + i) for i, p in enumerate(ps)]
        opt.zero_grad()
        for p, r, g in zip(gp, ref, grads):
            r.grad = g.clone()
            p.grad.copy_(g.to(dev))
        oref.step()
        opt.step()
        for p, r in zip(gp, ref):
            assert rel(p, r) < 1e-6


def test_layout_add_concat(dev):
    from cat_amd import ops
    x = detfill.normal((2, 5, 7, 9), 600)
    xg = ops.to_nhwc(x.to(dev))
    assert ops.is_act(xg) and torch.equal(xg.cpu(), x)
    assert torch.equal(ops.to_nchw(xg).cpu(), x)
    ys = [detfill.normal((2, 5, 7, 9), 601 + i) for i in range(7)]
    s = ops.AddNFn.apply(*[_nhwc(y, dev) for y in ys])
    assert rel(s, sum(ys)) < 1e-6
    a, b = detfill.normal((2, 3, 8, 8), 610), detfill.normal((2, 3, 8, 8), 611)
    ag, bg = _nhwc(a, dev, True), _nhwc(b, dev, True)
    c = ops.Concat2Fn.apply(ag, bg)
    assert torch.equal(c.cpu(), torch.cat((a, b), 1))
    gy = detfill.normal((2, 6, 8, 8), 612)
    c.backward(_nhwc(gy, dev))
    assert torch.equal(ag.grad.cpu(), gy[:, :3]) and torch.equal(bg.grad.cpu(), gy[:, 3:])
    # fan-out: gradients of the aliases are summed by one kernel
    xg = _nhwc(x, dev, True)
    outs = ops.fanout(xg, 10)
    gs = [detfill.normal((2, 5, 7, 9), 620 + i) for i in range(10)]
    torch.autograd.backward(list(outs), [_nhwc(g, dev) for g in gs])
    assert rel(xg.grad, sum(gs)) < 1e-6


def test_errors_are_loud(dev):
    from cat_amd import ops
    with pytest.raises(RuntimeError):
        ops.Conv2dFn.apply(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1, 0, 0, 0.0)   # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.Conv2dFn.apply(torch.zeros(1, 3, 8, 8, device=dev), torch.zeros(4, 5, 3, 3, device=dev), None, 1, 1, 0, 0, 0.0)


def test_integration_md_stub_runs_verbatim(dev):
    """The ctypes binding printed in INTEGRATION.md (what a CAT maintainer would paste) must work as written."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = [b for b in blocks if 'def conv2d_nhwc' in b]
    assert len(code) == 1
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec(code[0], ns)
        x = detfill.normal((2, 8, 12, 10), 71)                       # NCHW reference input, 8 channels (already a multiple of 4)
        w = detfill.normal((5, 8, 3, 3), 72, 0.2)
        b = detfill.normal((5,), 73, 0.1)
        y = ns['conv2d_nhwc'](x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev), 1, 1, True)
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, b)
    got = y[..., :5].permute(0, 3, 1, 2).cpu()
    assert rel(got, ref) < TOL
    assert float(y[..., 5:].abs().max()) == 0.0                      # padding channels are written as zeros


@pytest.mark.parametrize('pad,n,c,h,w', [(1, 2, 11, 9, 7), (2, 1, 42, 6, 10), (2, 3, 4, 1, 5)])
def test_replication_pad(dev, pad, n, c, h, w):
    """padding_type='replicate' (reference inception_modules.py:114-115): forward copy and backward fold against F.pad."""
    from cat_amd import ops
    x = detfill.normal((n, c, h, w), 301)
    xr = x.clone().requires_grad_(True)
    yr = F.pad(xr, (pad, pad, pad, pad), mode='replicate')
    gy = detfill.normal(tuple(yr.shape), 302)
    yr.backward(gy)
    xg = _nhwc(x, dev, True)
    y = ops.ReplicatePadFn.apply(xg, pad)
    assert tuple(y.shape) == tuple(yr.shape) and torch.equal(y.detach().cpu(), yr.detach())
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < 1e-6


def test_block_with_replicate_padding_and_relu6(dev):
    """InvertedResidualChannels(padding_type='replicate', active_fn=nn.ReLU6) -- two reference options no launch script selects -- run
    the general per-layer path and match stock torch (forward and input gradient)."""
    import functools
    from torch import nn
    from cat_amd import fused_block, nn as cnn, ops
    from cat_amd.inception_modules import InvertedResidualChannels, get_active_fn
    C, res, dw = 20, [6, 0, 5], [4, 7, 0]
    blk = InvertedResidualChannels(C, res, dw, 1, [1, 3, 5], [1, 3, 5], padding_type='replicate', use_bias=False, norm_layer=cnn.BatchNorm2d,
                                   norm_kwargs={'momentum': 0.1, 'eps': 1e-5}, active_fn=get_active_fn('nn.ReLU6'))
    sd = detfill.fill_state_dict(blk.state_dict(), 311, gamma_abs_normal=True)
    for k in sd:
        if k.endswith('.weight') and sd[k].dim() == 1:
            sd[k] = sd[k] * 4.0          # wide pre-activations: both bounds of ReLU6 are hit
    blk.load_state_dict(sd)
    blk = blk.to(dev).train()
    x = detfill.normal((2, C, 24, 20), 312)
    assert not fused_block.applicable(blk, ops.to_nhwc(x.to(dev)))
    mk = lambda c: nn.BatchNorm2d(c, momentum=0.1, eps=1e-5)
    res_ops, dw_ops = nn.ModuleList(), nn.ModuleList()
    for m, k in zip(res, [1, 3, 5]):
        if m:
            res_ops.append(nn.Sequential(nn.ReplicationPad2d((k - 1) // 2), nn.Sequential(nn.Conv2d(C, m, k, bias=False), mk(m), nn.ReLU6()), nn.Dropout(0.0),
                                         nn.ReplicationPad2d((k - 1) // 2), nn.Conv2d(m, C, k, bias=False)))
    for m, k in zip(dw, [1, 3, 5]):
        if m:
            dw_ops.append(nn.Sequential(nn.Sequential(nn.Conv2d(C, m, 1, bias=False), mk(m), nn.ReLU6()), nn.ReplicationPad2d((k - 1) // 2),
                                        nn.Sequential(nn.Conv2d(m, m, k, groups=m, bias=False), mk(m), nn.ReLU6()), nn.Dropout(0.0),
                                        nn.Conv2d(m, C, 1, bias=False)))
    twin = nn.Module()
    twin.res_ops, twin.dw_ops, twin.pw_bn = res_ops, dw_ops, mk(C)
    twin.load_state_dict({k: v.clone() for k, v in sd.items()})
    twin.train()
    xr = x.clone().requires_grad_(True)
    yr = xr + twin.pw_bn(sum(op(xr) for op in twin.res_ops) + sum(op(xr) for op in twin.dw_ops))
    gy = detfill.normal(tuple(yr.shape), 313)
    yr.backward(gy)
    xg = _nhwc(x, dev, True)
    y = blk(xg)
    assert rel(y, yr) < 1e-4
    y.backward(_nhwc(gy, dev))
    assert rel(xg.grad, xr.grad) < 5e-4


def test_wgrad_batch_equals_separate_calls(dev):
    """cat_conv2d_wgrad_batch (ops.WgradBatch): several weight gradients with ONE reduce launch for their partial sums must be bit-identical to
    one cat_conv2d_wgrad call each -- same producer kernels, same summation order (the fused block's seven narrow layers; here one layer per
    producer family: 1 x 1 pixel-split, LDS-tile 5 x 5 and 3 x 3, the generic split, and a direct single-slice launch), with and without
    accumulation into an existing gradient."""
    import ctypes as C
    from cat_amd import _lib as L, ops
    g = torch.Generator().manual_seed(5)
    n, h, w = 4, 32, 32
    layers = [(77, 15, 1, 0), (77, 18, 5, 1), (77, 12, 3, 1), (40, 130, 3, 0), (8, 8, 3, 0)]      # Cin, Cout, k, reflect
    xs, dys, dsts_a, dsts_b, geoms = [], [], [], [], []
    for cin, cout, k, refl in layers:
        x = ops.to_nhwc(torch.randn(n, cin, h, w, generator=g).to(dev))
        dy = ops.to_nhwc(torch.randn(n, cout, h, w, generator=g).to(dev))
        wa = ops.padded_weight_like((cout, cin, k, k), dev)
        wa.copy_(torch.randn(cout, cin, k, k, generator=g).to(dev))          # the gradient already in the buffer (accumulate = 1 for odd layers)
        wb = ops.padded_weight_like((cout, cin, k, k), dev)
        wb.copy_(wa)
        geoms.append(ops._conv_geom(n, h, w, cin, ops.act_cs(x), h, w, cout, ops.act_cs(dy), k, k, 1, k // 2, L.PAD_REFLECT if refl else L.PAD_ZERO,
                                    wcs=ops._grad_wcs(wa)))
        xs.append(x); dys.append(dy); dsts_a.append(wa); dsts_b.append(wb)
    for i, gw in enumerate(geoms):
        ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
        L.call('cat_conv2d_wgrad', C.byref(gw), ops._p(xs[i]), ops._p(dys[i]), ops._p(dsts_a[i]), i & 1, ops._p(ws), ops._stream())
    batch = ops.WgradBatch(dev, {})
    for i, gw in enumerate(geoms):
        batch.add_into(dsts_b[i], i & 1, gw, ops._p(xs[i]), ops._p(dys[i]))
    batch.flush()
    torch.cuda.synchronize()
    for a, b, lay in zip(dsts_a, dsts_b, layers):
        assert torch.equal(torch.as_strided(a, (a.untyped_storage().nbytes() // 4,), (1,), 0), torch.as_strided(b, (b.untyped_storage().nbytes() // 4,), (1,), 0)), lay
    # and against ATen for the one written fresh with zero padding (even index: accumulate = 0)
    want = torch.nn.functional.conv2d(ops.to_nchw(xs[4]).cpu().transpose(0, 1), ops.to_nchw(dys[4]).cpu().transpose(0, 1), padding=1).transpose(0, 1)
    got = dsts_b[4].cpu()
    assert float((got.double() - want.double()).abs().max() / want.double().abs().max()) < 1e-4


@pytest.mark.parametrize('c,h,w', [(3, 37, 53), (1, 8, 8), (4, 16, 20), (6, 9, 9), (70, 5, 7)])
def test_nchw_to_nhwc_layouts(dev, c, h, w):
    """cat_nchw_to_nhwc: the C <= 4 pixel-per-lane form (RGB batches of set_input, pixel stride 4, padding channel zero) and the tiled transpose,
    against a permute on the host; round trip through cat_nhwc_to_nchw."""
    from cat_amd import ops
    x = torch.randn(3, c, h, w, generator=torch.Generator().manual_seed(c * 100 + h))
    y = ops.to_nhwc(x.to(dev))
    cs = ops.act_cs(y)
    raw = torch.as_strided(y, (3, h, w, cs), (h * w * cs, w * cs, cs, 1)).cpu()
    assert torch.equal(raw[..., :c], x.permute(0, 2, 3, 1))
    if cs > c:
        assert float(raw[..., c:].abs().max()) == 0.0
    assert torch.equal(ops.to_nchw(y).cpu(), x)
