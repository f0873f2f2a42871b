"""GPU parity of the K-concatenated wide-tile convolution sum (csrc/conv_ksum.hip, cat_conv2d_ksum_fwd through ctypes) against stock fp32
ATen on the host: y = act(bias + sum_s conv_{k x k, same}(src_s, w_s)) + res -- the frozen teacher's block tail (reference
models/modules/inception_modules.py:230-236 with eval-mode BatchNorm folded).  Tolerance 1e-4 of the output's range (bar: 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    from cat_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _ref(srcs, ws, refl, bias, res, act):
    y = None
    for x, w in zip(srcs, ws):
        k = w.shape[2]
        p = k // 2
        xp = F.pad(x, (p, p, p, p), mode='reflect') if (refl and p) else x
        t = F.conv2d(xp, w, None, 1, 0 if (refl and p) else p)
        y = t if y is None else y + t
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if act == 1:
        y = F.relu(y)
    if res is not None:
        y = y + res
    return y


CASES = [
    # (channels, kernel) per segment, Cout, reflect, N, H, W, bias, res, act
    ([(176, 1), (42, 3), (42, 5)], 256, True, 2, 16, 16, True, True, 0),      # the teacher's tail in small: 6 + 5 quad chunks, 8/8/7/7/7/7
    ([(176, 1), (42, 3), (42, 5)], 256, False, 1, 12, 20, True, True, 0),     # zero padding: out-of-plane rows / columns through the markers
    ([(44, 3)], 130, True, 3, 9, 11, False, False, 1),                        # ragged M (297 pixels) and N (130): tile tails, ReLU epilogue
    ([(8, 5), (20, 1), (4, 3), (36, 3)], 77, False, 2, 10, 10, True, False, 0),   # 1..3-quad remainder steps only / 2 full sets + 1
    ([(64, 3), (12, 1)], 128, True, 1, 8, 64, False, True, 0),                # 2 full sets per chunk (16 quads = 8 + 8), 3-quad chunk
    ([(256, 1)], 64, False, 2, 7, 9, True, True, 1),
    ([(256, 1)], 176, False, 2, 16, 16, True, False, 1),                      # the teacher's merged first 1 x 1 GEMM: two 96-wide N tiles
    ([(44, 3), (20, 1)], 150, True, 1, 12, 12, True, True, 0),                # 96-wide tiles with a ragged second tile (54 of 96 columns)
    ([(32, 5)], 96, False, 1, 9, 9, False, False, 0),                         # exactly one 96-wide tile
]


@pytest.mark.parametrize('segs,cout,refl,n,h,w,has_bias,has_res,act', CASES)
def test_ksum_matches_aten(dev, segs, cout, refl, n, h, w, has_bias, has_res, act):
    from cat_amd import ksum, ops
    g = torch.Generator().manual_seed(1234 + cout + n * h)
    srcs = [torch.randn(n, c, h, w, generator=g) for c, _ in segs]
    ws = [torch.randn(cout, c, k, k, generator=g) / (c * k * k) ** 0.5 for c, k in segs]
    bias = torch.randn(cout, generator=g) if has_bias else None
    res = torch.randn(n, cout, h, w, generator=g) if has_res else None
    want = _ref(srcs, ws, refl, bias, res, act)
    ks = []
    for x, wt in zip(srcs, ws):
        wd = ops.padded_weight_like(tuple(wt.shape), dev)
        wd.copy_(wt.to(dev))
        ks.append(ksum.Segment(ops.to_nhwc(x.to(dev)), wd, refl))
    old = ksum.set_min_workgroups(1)
    try:
        assert ksum.applicable(ks, n, h, w, cout)
    finally:
        ksum.set_min_workgroups(old)
    y = ops.empty_act(n, cout, h, w, dev)
    y.fill_(float('nan'))      # every valid element must be written; padding channels must come out as exact zeros
    ksum.run(ks, None if bias is None else bias.to(dev), y, res=None if res is None else ops.to_nhwc(res.to(dev)), act=act)
    torch.cuda.synchronize()
    got = y.detach().cpu()
    err = float((got.double() - want.double()).abs().max() / (want.double().abs().max() + 1e-12))
    assert err < TOL, err
    cs = ops.act_cs(y)
    if cs > cout:
        raw = torch.as_strided(y, (n, h, w, cs), (h * w * cs, w * cs, cs, 1)).cpu()
        assert float(raw[..., cout:].abs().max()) == 0.0


def test_ksum_channel_slice_sources(dev):
    """Segments that are channel SLICES of one wider buffer (pixel stride 176, 44-channel slices): what a concatenated hidden tensor looks like."""
    from cat_amd import ksum, ops
    g = torch.Generator().manual_seed(7)
    n, h, w, cout = 2, 8, 16, 96
    wide = ops.to_nhwc(torch.randn(n, 176, h, w, generator=g).to(dev))
    wts = [torch.randn(cout, 44, k, k, generator=g) / (44 * k * k) ** 0.5 for k in (1, 3, 5, 3)]
    ks, srcs = [], []
    for i, wt in enumerate(wts):
        wd = ops.padded_weight_like(tuple(wt.shape), dev)
        wd.copy_(wt.to(dev))
        sl = wide[:, 44 * i:44 * (i + 1)]
        assert ops.is_act(sl)
        ks.append(ksum.Segment(sl, wd, True))
        srcs.append(sl.detach().cpu().contiguous())
    want = _ref(srcs, wts, True, None, None, 0)
    old = ksum.set_min_workgroups(1)
    try:
        assert ksum.applicable(ks, n, h, w, cout)
    finally:
        ksum.set_min_workgroups(old)
    y = ops.empty_act(n, cout, h, w, dev)
    ksum.run(ks, None, y)
    torch.cuda.synchronize()
    err = float((y.cpu().double() - want.double()).abs().max() / want.double().abs().max())
    assert err < TOL, err


@pytest.mark.parametrize('refl,n,h,w,cin', [(True, 2, 16, 32, 256), (False, 1, 13, 21, 256), (True, 3, 8, 16, 40), (False, 2, 9, 16, 22)])
def test_teacher_stage1_one_launch_matches_aten(dev, refl, n, h, w, cin):
    """cat_tstage1w_fwd: the frozen teacher block's first convs -- 5 x 5 and 3 x 3 (-> 42) and the concatenated 1 x 1 convs (-> 176, four
    42-channel slices in 44-channel slots) -- with bias + ReLU from one staging of x (inception_modules.py:135-165 with eval BatchNorm folded),
    against F.conv2d per conv on the host.  Ragged planes, zero / mirrored borders, a channel count that is not a multiple of 16 or of 4."""
    import ctypes as C
    from cat_amd import _lib as L, ops, tconv
    g = torch.Generator().manual_seed(99 + h * w + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    w5 = torch.randn(42, cin, 5, 5, generator=g) / (cin * 25) ** 0.5
    w3 = torch.randn(42, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    w1 = torch.zeros(176, cin, 1, 1)
    b1 = torch.zeros(176)
    for s in range(4):
        w1[44 * s:44 * s + 42] = torch.randn(42, cin, 1, 1, generator=g) / cin ** 0.5
        b1[44 * s:44 * s + 42] = torch.randn(42, generator=g)
    b5, b3 = torch.randn(42, generator=g), torch.randn(42, generator=g)
    want = [_ref([x], [wt], refl, b, None, 1) for wt, b in ((w5, b5), (w3, b3), (w1, b1))]
    assert L.query('cat_tstage1w_supported', 42, 42, 176)
    packs, keep = [], []
    for wt in (w5, w3, w1):
        wd = ops.padded_weight_like(tuple(wt.shape), dev)
        wd.copy_(wt.to(dev))
        keep.append(wd)
        packs.append(tconv.pack(wd, tconv.FWD))
    xd = ops.to_nhwc(x.to(dev))
    biases = [b.to(dev) for b in (b5, b3, b1)]
    ys = [ops.empty_act(n, m, h, w, dev) for m in (42, 42, 176)]
    for y in ys:
        y.fill_(float('nan'))
    geo = L.Stage1WGeom()
    geo.N, geo.H, geo.W, geo.xcs, geo.cin, geo.reflect, geo.act, geo.slope = n, h, w, ops.act_cs(xd), cin, int(refl), L.ACT_RELU, 0.0
    for k, m in enumerate((42, 42, 176)):
        geo.ycs[k], geo.nvalid[k] = ops.act_cs(ys[k]), m
    arr = C.c_void_p * 3
    L.call('cat_tstage1w_fwd', C.byref(geo), ops._p(xd), arr(*[t.data_ptr() for t in packs]), arr(*[t.data_ptr() for t in biases]),
           arr(*[t.data_ptr() for t in ys]), ops._stream())
    torch.cuda.synchronize()
    for y, wnt, m in zip(ys, want, (42, 42, 176)):
        err = float((y.cpu().double() - wnt.double()).abs().max() / wnt.double().abs().max())
        assert err < TOL, err
        cs = ops.act_cs(y)
        raw = torch.as_strided(y, (n, h, w, cs), (h * w * cs, w * cs, cs, 1)).cpu()
        assert not torch.isnan(raw).any()
        if cs > m:
            assert float(raw[..., m:].abs().max()) == 0.0
