"""GPU parity of the quad-granule LDS-tile convolution (csrc/conv_q.hip: v_mfma_f32_4x4x1_16B with broadcast filters) against stock fp32
ATen on the host, called through the C-ABI (cat_amd.qconv -> ctypes -> cat_qconv_plan / _pack / _fwd, cat_tnorm_finalize2).  The layers are
those of the generator's edge (reference models/modules/inception_architecture/inception_generator.py:37-56,118-132) at ragged student
widths, plus stride-1 layers of the blocks.  Tolerance 1e-4 relative to the tensor's largest entry (north_star: 1e-3); observed ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

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


def _gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _weight(w, dev):
    from cat_amd import ops
    out = ops.padded_weight_like(w.shape, dev) if w.shape[1] > 1 else torch.empty(w.shape, device=dev)
    out.copy_(w.to(dev))
    return out


CONV_CASES = [
    # cin, cout, k, stride, reflect, act, bias, N, H, W
    (3, 22, 7, 1, True, 0, False, 2, 40, 48),          # image stem (ck 4, 4 taps per filter register)
    (3, 16, 7, 1, True, 1, True, 5, 64, 64),           # ... 16 x 16 tiles (4 pixel groups)
    (16, 3, 7, 1, True, 3, True, 2, 37, 45),           # tanh head, ragged plane
    (16, 3, 7, 1, True, 3, True, 5, 64, 64),
    (22, 37, 3, 2, False, 0, False, 2, 40, 48),        # stride 2: parity-split patch, ck 8
    (37, 77, 3, 2, False, 0, True, 3, 33, 31),         # odd planes
    (4, 8, 3, 2, False, 0, False, 1, 16, 16),
    (77, 18, 5, 1, True, 1, False, 2, 24, 32),         # block layers
    (18, 77, 5, 1, True, 0, True, 2, 24, 32),
    (77, 12, 3, 1, False, 0, False, 2, 17, 19),
    (40, 16, 1, 1, False, 0, False, 2, 16, 32),        # c4 = 40: 8-channel chunks
    (256, 42, 5, 1, True, 0, False, 1, 16, 16),        # teacher widths
    (42, 256, 3, 1, True, 0, False, 1, 16, 16),        # 64 output quads: N blocks over the grid
    (24, 200, 3, 1, False, 2, True, 1, 12, 20),
]


@pytest.fixture(params=[1024, 1], ids=['tiles8x16', 'tiles16x16'])
def tiling(request):
    """Both tilings of narrow outputs: the default rule needs >= 1024 16 x 16 tiles before it picks them (batch 16 at 256 x 256)."""
    from cat_amd import _lib as L
    old = L.query('cat_qconv_min_tiles16', request.param)
    yield request.param
    L.query('cat_qconv_min_tiles16', old)


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_qconv_matches_torch(dev, case, tiling):
    from cat_amd import _lib as L, ops, qconv
    cin, cout, k, stride, reflect, act, bias, n, h, w = case
    pad = (k - 1) // 2
    x = _gen(n, cin, h, w, seed=1)
    wt = _gen(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    b = _gen(cout, seed=3) if bias else None
    xp = F.pad(x, (pad,) * 4, mode='reflect') if reflect and pad else x
    ref = F.conv2d(xp, wt, b, stride=stride, padding=0 if reflect else pad)
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.2)
    elif act == 3:
        ref = torch.tanh(ref)
    xd = ops.to_nhwc(x.to(dev))
    layer = qconv.Layer('conv', _weight(wt, dev), stride=stride, pad=pad, reflect=reflect)
    assert qconv.Layer.supported('conv', wt, stride, pad)
    cout_, ho, wo = layer.out_shape(xd)
    y = ops.empty_act(n, cout, ho, wo, dev)
    full = torch.as_strided(y, (n, ops.act_cs(y), ho, wo), y.stride())
    full.fill_(7.0)                                     # padding channels must come out as 0
    layer.run(xd, None if b is None else b.to(dev), y, act=act, slope=0.2)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel(y, ref) < TOL
    if ops.act_cs(y) > cout:
        assert float(full[:, cout:].abs().max()) == 0.0
    # a second call re-uses plan and packed filters
    layer.run(xd, None if b is None else b.to(dev), y, act=act, slope=0.2)
    assert rel(y, ref) < TOL


CT_CASES = [(77, 38, 2, 16, 16), (38, 16, 2, 32, 32), (38, 16, 5, 64, 64), (82, 38, 1, 9, 11), (25, 12, 3, 13, 7)]


@pytest.mark.parametrize('case', CT_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_qconv_transposed_matches_torch(dev, case, tiling):
    from cat_amd import ops, qconv
    cin, cout, n, h, w = case
    x = _gen(n, cin, h, w, seed=4)
    wt = _gen(cin, cout, 3, 3, seed=5, scale=(cin * 9) ** -0.5)
    b = _gen(cout, seed=6)
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
    xd = ops.to_nhwc(x.to(dev))
    layer = qconv.Layer('convt', _weight(wt, dev), stride=2, pad=1)
    y = ops.empty_act(n, cout, 2 * h, 2 * w, dev)
    layer.run(xd, b.to(dev), y)
    torch.cuda.synchronize()
    assert rel(y, ref) < TOL


def test_qconv_staging_affine_and_activation(dev):
    """The norm + ReLU in front of a conv applied while its input is staged: per-channel (BatchNorm) and per-image (InstanceNorm) rows;
    zero padding must stay zero AFTER the affine (a padded pixel is not `shift`)."""
    from cat_amd import _lib as L, ops, qconv
    n, cin, cout, h, w = 3, 22, 37, 20, 24
    x = _gen(n, cin, h, w, seed=7)
    wt = _gen(cout, cin, 3, 3, seed=8, scale=0.1)
    xd = ops.to_nhwc(x.to(dev))
    layer = qconv.Layer('conv', _weight(wt, dev), stride=2, pad=1)
    for per_image in (False, True):
        rows = n if per_image else 1
        sc, sh = _gen(rows, cin, seed=9) * 0.5 + 1.0, _gen(rows, cin, seed=10)
        a = F.relu(x * sc.view(rows, cin, 1, 1) + sh.view(rows, cin, 1, 1))
        ref = F.conv2d(a, wt, None, stride=2, padding=1)
        c4 = qconv.cs4(cin)
        scd, shd = torch.zeros(rows, c4, device=dev), torch.zeros(rows, c4, device=dev)
        scd[:, :cin], shd[:, :cin] = sc.to(dev), sh.to(dev)
        y = ops.empty_act(n, cout, 10, 12, dev)
        layer.run(xd, None, y, pre=(scd, shd, c4 if per_image else 0, L.ACT_RELU, 0.0))
        torch.cuda.synchronize()
        assert rel(y, ref) < TOL


@pytest.mark.parametrize('kind', ['conv7', 'convs2', 'convt', 'conv5'])
@pytest.mark.parametrize('groups', ['batch', 'instance'])
def test_qconv_tile_statistics_and_finalize(dev, kind, groups, tiling):
    """Per-tile (sum, M2) from the conv epilogue + cat_tnorm_finalize2 = the statistics of nn.BatchNorm2d / nn.InstanceNorm2d in training
    mode (biased variance in rstd, unbiased in running_var), including ragged edge tiles and the four classes of a transposed conv."""
    import ctypes as C
    from cat_amd import _lib as L, ops, qconv
    n = 3
    if kind == 'conv7':
        cin, cout, h, w = 3, 22, 37, 45
        wt = _gen(cout, cin, 7, 7, seed=11, scale=0.1)
        layer = qconv.Layer('conv', _weight(wt, dev), stride=1, pad=3, reflect=True)
        x = _gen(n, cin, h, w, seed=12)
        z = F.conv2d(F.pad(x, (3,) * 4, mode='reflect'), wt)
    elif kind == 'convs2':
        cin, cout, h, w = 22, 37, 37, 45
        wt = _gen(cout, cin, 3, 3, seed=11, scale=0.1)
        layer = qconv.Layer('conv', _weight(wt, dev), stride=2, pad=1)
        x = _gen(n, cin, h, w, seed=12)
        z = F.conv2d(x, wt, stride=2, padding=1)
    elif kind == 'conv5':
        cin, cout, h, w = 77, 18, 20, 24
        wt = _gen(cout, cin, 5, 5, seed=11, scale=0.05)
        layer = qconv.Layer('conv', _weight(wt, dev), stride=1, pad=2, reflect=True)
        x = _gen(n, cin, h, w, seed=12)
        z = F.conv2d(F.pad(x, (2,) * 4, mode='reflect'), wt)
    else:
        cin, cout, h, w = 38, 16, 19, 21
        wt = _gen(cin, cout, 3, 3, seed=11, scale=0.1)
        layer = qconv.Layer('convt', _weight(wt, dev), stride=2, pad=1)
        x = _gen(n, cin, h, w, seed=12)
        z = F.conv_transpose2d(x, wt, stride=2, padding=1, output_padding=1)
    bias = _gen(cout, seed=13)
    z = z + bias.view(1, -1, 1, 1)
    xd = ops.to_nhwc(x.to(dev))
    _, ho, wo = layer.out_shape(xd)
    y = ops.empty_act(n, cout, ho, wo, dev)
    scs = qconv.cs4(cout)
    stats = torch.full((n * 4096 * 2 * scs,), float('nan'), device=dev)      # generous; the plan says how many entries are used
    plan = layer.run(xd, bias.to(dev), y, stats=stats, scs=scs)
    torch.cuda.synchronize()
    assert rel(y, z) < TOL
    used = n * plan.tiles * 2 * scs
    assert not torch.isnan(stats[:used]).any() and torch.isnan(stats[used:]).all()
    G = n if groups == 'instance' else 1
    gamma, beta = (_gen(cout, seed=14) * 0.3 + 1.0), _gen(cout, seed=15)
    gd, bd = torch.zeros(scs, device=dev), torch.zeros(scs, device=dev)
    gd[:cout], bd[:cout] = gamma.to(dev), beta.to(dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    sl = (L.NSlice * 1)()
    sl[0].c0, sl[0].c = 0, cout
    if groups == 'batch':
        sl[0].running_mean, sl[0].running_var = rm.data_ptr(), rv.data_ptr()
    scale, shift = torch.empty(G, scs, device=dev), torch.empty(G, scs, device=dev)
    mean, rstd = torch.empty(G, scs, device=dev), torch.empty(G, scs, device=dev)
    ncls = 4 if kind == 'convt' else 1
    lat = (h, w) if kind == 'convt' else (ho, wo)
    L.call('cat_tnorm_finalize2', ops._p(stats), scs, G, n, lat[0], lat[1], plan.th, plan.tw, ncls, ops._p(gd), ops._p(bd), 1, sl, 1e-5, 0.1,
           ops._p(scale), ops._p(shift), ops._p(mean), ops._p(rstd), scs, ops._stream())
    torch.cuda.synchronize()
    dims = (2, 3) if groups == 'instance' else (0, 2, 3)
    m = z.mean(dims, keepdim=False).reshape(G, cout)
    v = z.var(dims, unbiased=False).reshape(G, cout)
    assert rel(mean[:, :cout], m) < TOL
    assert rel(rstd[:, :cout], (v + 1e-5).rsqrt()) < TOL
    sc_ref = gamma.view(1, -1) * (v + 1e-5).rsqrt()
    assert rel(scale[:, :cout], sc_ref) < TOL
    assert rel(shift[:, :cout], beta.view(1, -1) - m * sc_ref) < TOL
    if groups == 'batch':
        cnt = z.numel() / cout
        assert rel(rm, 0.1 * m[0]) < TOL
        assert rel(rv, 0.9 + 0.1 * v[0] * cnt / (cnt - 1)) < TOL


def test_qconv_k_segments_sum(dev):
    """Several convolutions of different kernel sizes over different tensors accumulated into one output (the branch sum of
    InvertedResidualChannels.forward, inception_modules.py:230-236) + residual."""
    from cat_amd import _lib as L, ops, qconv
    n, h, w, cout = 2, 16, 32, 30
    specs = [(18, 5), (12, 3), (9, 1)]
    xs = [_gen(n, c, h, w, seed=20 + i) for i, (c, k) in enumerate(specs)]
    ws = [_gen(cout, c, k, k, seed=30 + i, scale=0.1) for i, (c, k) in enumerate(specs)]
    res = _gen(n, cout, h, w, seed=40)
    ref = res.clone()
    for x, wt, (c, k) in zip(xs, ws, specs):
        ref = ref + F.conv2d(x, wt, padding=(k - 1) // 2)
    xd = [ops.to_nhwc(x.to(dev)) for x in xs]
    segs = [qconv.Seg(x, k, k, -((k - 1) // 2), -((k - 1) // 2)) for x, (c, k) in zip(xd, specs)]
    resd = ops.to_nhwc(res.to(dev))
    y = ops.empty_act(n, cout, h, w, dev)
    g = qconv.geometry(segs, n, h, w, h, w, cout, ops.act_cs(y), res=resd)
    plan = qconv.plan_of(g)
    buf = torch.zeros(int(plan.pack_floats), device=dev)
    for s, (wt, (c, k)) in enumerate(zip(ws, specs)):
        wd = _weight(wt, dev)
        wcl, wcs = ops.weight_cl(wd)
        qconv.pack_conv(g, s, buf, wcl, wcs, cout, k * k)
    qconv.launch(g, buf, None, y.data_ptr())
    torch.cuda.synchronize()
    assert rel(y, ref) < TOL


@pytest.mark.parametrize('norm', ['batch', 'instance'])
def test_generator_edge_through_fused_sequential(dev, norm):
    """ReflectionPad2d(3) . Conv 7x7 . norm . ReLU . Conv 3x3 / 2 . norm . ReLU . Conv 3x3 / 2 (wide) . norm . ReLU -- the down-sampling
    head of InceptionGenerator (reference inception_generator.py:37-56) with a pruned student's widths -- through cat_amd.nn.FusedSequential:
    the three convs take the quad-granule kernel with statistics in the epilogue.  No-grad forward (pending norms applied in the next
    conv's staging), grad-mode forward, every gradient and the running statistics against stock torch on the host."""
    from torch import nn as tnn
    from cat_amd import nn as cnn, ops, qconv
    old = ops.set_tconv_min_tiles(1)
    try:
        torch.manual_seed(5)
        inst = norm == 'instance'
        widths = (3, 22, 37, 77)

        def build(mod):
            Norm = (lambda c: mod.InstanceNorm2d(c, affine=True)) if inst else (lambda c: mod.BatchNorm2d(c))
            layers = [mod.ReflectionPad2d(3), mod.Conv2d(widths[0], widths[1], 7, padding=0, bias=inst), Norm(widths[1]), mod.ReLU(True)]
            for a, b in zip(widths[1:-1], widths[2:]):
                layers += [mod.Conv2d(a, b, 3, stride=2, padding=1, bias=inst), Norm(b), mod.ReLU(True)]
            return layers

        ref = tnn.Sequential(*build(tnn))
        for p_ in ref.parameters():
            p_.data.normal_(0, 0.3)
        net = cnn.FusedSequential(*build(cnn)).to(dev)
        net.load_state_dict(ref.state_dict())
        ref.train(), net.train()
        x = _gen(2, 3, 40, 48, seed=21)
        xd = ops.to_nhwc(x.to(dev))
        calls = {'n': 0}
        orig = qconv.Layer.run

        def counting(self, *a, **k):
            calls['n'] += 1
            return orig(self, *a, **k)
        qconv.Layer.run = counting
        try:
            with torch.no_grad():
                y0 = net(xd)
            assert calls['n'] == 3, 'the stem and the two stride-2 convs take the quad-granule kernel'
            xg = xd.detach().requires_grad_(True)
            y1 = net(xg)
            assert calls['n'] == 6
        finally:
            qconv.Layer.run = orig
        xr = x.clone().requires_grad_(True)
        ref2 = tnn.Sequential(*build(tnn))
        ref2.load_state_dict(ref.state_dict())
        ref2.train()
        with torch.no_grad():
            yr0 = ref2(x)                      # first training-mode forward (running statistics move once, like net's no-grad pass)
        yr = ref2(xr)
        assert rel(y0, yr0) < TOL and rel(y1, yr) < TOL
        gy = _gen(*yr.shape, seed=22)
        yr.backward(gy)
        y1.backward(ops.to_nhwc(gy.to(dev)))
        torch.cuda.synchronize()
        assert rel(xg.grad, xr.grad) < 5e-4
        gmax = max(float(pr.grad.abs().max()) for pr in ref2.parameters())
        for (k, pg), (_, pr) in zip(net.named_parameters(), ref2.named_parameters()):
            if float(pr.grad.abs().max()) < 1e-4 * gmax:      # a conv bias in front of a norm: the exact gradient is 0, both sides hold round-off
                assert float((pg.grad.cpu() - pr.grad).abs().max()) < 1e-3 * gmax, k
            else:
                assert rel(pg.grad, pr.grad) < 5e-4, k
        if not inst:
            for (k, bg), (_, br) in zip(net.named_buffers(), ref2.named_buffers()):
                assert rel(bg.float(), br.float()) < TOL, k
    finally:
        ops.set_tconv_min_tiles(old)


def test_hooked_edge_layers_keep_their_hooks(dev):
    """Forward (pre-)hooks on the generator's edge layers must keep firing with tensor arguments (the distillers tap `down_sampling.9` this way,
    base_inception_distiller.py:247-264): a hooked conv does not take the quad-granule path (its forward would never be called), a hooked
    ReflectionPad2d never receives a pending (unwritten) norm.  Output equals the un-hooked run."""
    from cat_amd import nn as cnn, ops, qconv
    old = ops.set_tconv_min_tiles(1)
    try:
        torch.manual_seed(6)

        def build():
            return cnn.FusedSequential(cnn.ReflectionPad2d(3), cnn.Conv2d(3, 22, 7, padding=0, bias=False), cnn.BatchNorm2d(22), cnn.ReLU(True),
                                       cnn.ReflectionPad2d(1), cnn.Conv2d(22, 37, 3, padding=0, bias=False), cnn.BatchNorm2d(37), cnn.ReLU(True),
                                       cnn.Conv2d(37, 40, 3, stride=2, padding=1, bias=False), cnn.BatchNorm2d(40), cnn.ReLU(True)).to(dev).train()
        plain = build()
        sd = {k: v.clone() for k, v in plain.state_dict().items()}
        x = ops.to_nhwc(_gen(2, 3, 40, 48, seed=31).to(dev))
        runs = {'n': 0}
        orig = qconv.Layer.run

        def counting(self, *a, **k):
            runs['n'] += 1
            return orig(self, *a, **k)
        qconv.Layer.run = counting
        try:
            with torch.no_grad():
                y_plain = plain(x)
            assert runs['n'] == 3
            for hooked_idx, expect_q in ((5, 2), (4, 3), (8, 2)):      # the second conv; the pad in front of it; the stride-2 conv
                net = build()
                net.load_state_dict(sd)
                seen = []
                net[hooked_idx].register_forward_pre_hook(lambda m, a: seen.append(('pre', type(a[0]))))
                net[hooked_idx].register_forward_hook(lambda m, a, o: seen.append(('post', type(o))))
                runs['n'] = 0
                with torch.no_grad():
                    y = net(x)
                assert [s[0] for s in seen] == ['pre', 'post'], (hooked_idx, seen)
                assert seen[0][1] is not ops.Normed, (hooked_idx, seen)          # never a pending norm
                assert runs['n'] == expect_q, (hooked_idx, runs['n'])
                assert rel(y, y_plain) < TOL, hooked_idx
        finally:
            qconv.Layer.run = orig
    finally:
        ops.set_tconv_min_tiles(old)
