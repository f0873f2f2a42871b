"""GPU check + micro-bench of the packed-filter LDS-tile convolution (csrc/conv_pk.hip) against ATen on the host.
  python tools/debug/check_conv_pk.py            correctness (single conv fwd, dgrad, K-concatenated multi-source with staging affine)
  python tools/debug/check_conv_pk.py --bench    the layers it targets, next to the im2col kernels (CAT_PK_TW=16|32 to force a tile)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cat_amd import _lib as L, ops, synthetic, tconv  # noqa: E402

dev = torch.device('cuda:0')


def conv_ref(x, w, b, k, reflect):
    p = (k - 1) // 2
    xp = F.pad(x, (p,) * 4, mode='reflect') if (reflect and p) else x
    return F.conv2d(xp, w, b, padding=0 if reflect else p)


def run_single(x, w, b, reflect, act, tw=None):
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    xg = ops.to_nhwc(x.to(dev))
    wg = ops.padded_weight_like(w.shape, dev)
    wg.copy_(w)
    pk = tconv.pack(wg, tconv.FWD)
    y = ops.empty_act(n, cout, h, wd, dev)
    seg = tconv.Segment(xg, k, (k - 1) // 2, reflect, 0)
    tconv.run([seg], pk, None if b is None else b.to(dev), y, cout, n, h, wd, h, wd, act, 0.2)
    return y


def check():
    worst = 0.0
    cases = [  # cin, cout, k, reflect, act, n, h, w
        (54, 7, 5, True, 0, 2, 16, 16), (96, 16, 3, True, 1, 1, 6, 6), (22, 18, 5, True, 0, 1, 9, 35), (10, 7, 3, False, 2, 2, 8, 32),
        (36, 42, 5, False, 1, 1, 11, 40), (16, 16, 3, True, 0, 1, 16, 33), (82, 17, 5, True, 1, 2, 24, 40), (256, 42, 3, True, 1, 1, 16, 64),
        (4, 1, 3, False, 0, 1, 5, 5), (77, 33, 5, False, 3, 1, 8, 8), (18, 77, 5, True, 0, 2, 16, 32), (12, 77, 3, True, 0, 1, 17, 19),
        (42, 256, 5, True, 0, 1, 16, 32), (9, 77, 5, False, 0, 1, 8, 16), (77, 53, 1, False, 1, 2, 8, 24), (20, 200, 3, False, 0, 1, 9, 17),
    ]
    for cin, cout, k, reflect, act, n, h, w in cases:
        x = synthetic.normal((n, cin, h, w), 1)
        wt = synthetic.normal((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
        b = synthetic.normal((cout,), 3, 0.1)
        yr = conv_ref(x, wt, b, k, reflect)
        yr = {0: yr, 1: F.relu(yr), 2: F.leaky_relu(yr, 0.2), 3: torch.tanh(yr)}[act]
        y = run_single(x, wt, b, reflect, act)
        err = float((y.cpu() - yr).abs().max() / yr.abs().max())
        cs = ops.act_cs(y)
        full = torch.as_strided(y, (y.shape[0], cs, y.shape[2], y.shape[3]), y.stride())
        padz = float(full[:, cout:].abs().max()) if cs > cout else 0.0
        print(f'fwd {cin}->{cout} k{k} reflect={reflect} act={act} {n}x{h}x{w}: rel err {err:.2e}, pad lanes {padz}')
        worst = max(worst, err, padz)
    # dgrad: gradient w.r.t. the (padded, for reflect) input
    for cin, cout, k, reflect, n, h, w in [(18, 82, 5, True, 1, 9, 33), (12, 54, 3, False, 2, 8, 20), (77, 17, 5, False, 1, 10, 34), (77, 12, 3, True, 1, 16, 40),
                                           (80, 9, 5, True, 2, 8, 16)]:
        p = (k - 1) // 2
        x = synthetic.normal((n, cin, h, w), 5)
        wt = synthetic.normal((cout, cin, k, k), 6, 1.0 / np.sqrt(cin * k * k))
        gy = synthetic.normal((n, cout, h, w), 7)
        xr = x.clone().requires_grad_(True)
        xp = F.pad(xr, (p,) * 4, mode='reflect') if reflect else xr
        xp.retain_grad()
        F.conv2d(xp, wt, None, padding=0 if reflect else p).backward(gy)
        ref = xp.grad if reflect else xr.grad           # reflect: gradient of the PADDED plane (the caller folds it)
        wg = ops.padded_weight_like(wt.shape, dev)
        wg.copy_(wt)
        pk = tconv.pack(wg, tconv.DGRAD)
        dyg = ops.to_nhwc(gy.to(dev))
        ho, wo = (h + 2 * p, w + 2 * p) if reflect else (h, w)
        dx = ops.empty_act(n, cin, ho, wo, dev)
        seg = tconv.Segment(dyg, k, k - 1 - (0 if reflect else p), False, 0)
        tconv.run([seg], pk, None, dx, cin, n, h, w, ho, wo)
        err = float((dx.cpu() - ref).abs().max() / ref.abs().max())
        print(f'dgrad {cin}<-{cout} k{k} reflect={reflect} {n}x{h}x{w}: rel err {err:.2e}')
        worst = max(worst, err)
    # K-concatenated branch sum with staging affine + ReLU: sum_b conv_kb(relu(h_b * s_b + t_b)) over channel slices of one buffer
    for reflect in (True, False):
        n, h, w, cout = 2, 16, 24, 77
        ms, kss = [11, 12, 18, 15, 15, 12], [1, 3, 5, 1, 1, 1]
        offs = np.cumsum([0] + [tconv.cs4(m) for m in ms])
        hc = int(offs[-1])
        hbuf = torch.zeros(n, h, w, hc)
        ref = torch.zeros(n, cout, h, w)
        segs, packs, poff = [], [], 0
        sc_all, sh_all = torch.zeros(hc), torch.zeros(hc)
        hg = hbuf.to(dev)
        for bi, (m, k) in enumerate(zip(ms, kss)):
            hb = synthetic.normal((n, m, h, w), 20 + bi)
            sc = synthetic.normal((m,), 40 + bi).abs() + 0.5
            sh = synthetic.normal((m,), 60 + bi, 0.3)
            wt = synthetic.normal((cout, m, k, k), 80 + bi, 1.0 / np.sqrt(m * k * k * 6))
            a = F.relu(hb * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
            ref += conv_ref(a, wt, None, k, reflect)
            hbuf[..., offs[bi]:offs[bi] + m] = hb.permute(0, 2, 3, 1)
            sc_all[offs[bi]:offs[bi] + m] = sc
            sh_all[offs[bi]:offs[bi] + m] = sh
            wg = ops.padded_weight_like(wt.shape, dev)
            wg.copy_(wt)
            packs.append(tconv.pack(wg, tconv.FWD))
        hg = hbuf.to(dev)
        scg, shg = sc_all.to(dev), sh_all.to(dev)
        pk = torch.cat(packs)
        for bi, (m, k) in enumerate(zip(ms, kss)):
            o = int(offs[bi])
            segs.append(tconv.Segment(hg, k, (k - 1) // 2, reflect, poff, c4=tconv.cs4(m), scale=scg[o:], shift=shg[o:], act=L.ACT_RELU,
                                      xcs=hc, ptr=hg.data_ptr() + 4 * o))
            poff += packs[bi].numel()
        b = synthetic.normal((cout,), 99, 0.1)
        ref += b.view(1, -1, 1, 1)
        y = ops.empty_act(n, cout, h, w, dev)
        tconv.run(segs, pk, b.to(dev), y, cout, n, h, w, h, w)
        err = float((y.cpu() - ref).abs().max() / ref.abs().max())
        print(f'multi-source 6 segments reflect={reflect}: rel err {err:.2e}')
        worst = max(worst, err)
    torch.cuda.synchronize()
    ok = worst < 1e-4
    print('OK' if ok else 'FAILED')
    return 0 if ok else 1


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def bench():
    shapes = [('S 77->18 k5', 77, 18, 5), ('S 77->14 k5', 77, 14, 5), ('S 77->12 k3', 77, 12, 3), ('S 18->77 k5', 18, 77, 5), ('S 14->77 k5', 14, 77, 5),
              ('S 12->77 k3', 12, 77, 3), ('T 256->42 k5', 256, 42, 5), ('T 42->256 k5', 42, 256, 5), ('T 256->42 k3', 256, 42, 3), ('T 42->256 k3', 42, 256, 3)]
    n, h, w = 16, 64, 64
    for name, cin, cout, k in shapes:
        x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
        wt = ops.padded_weight_like((cout, cin, k, k), dev)
        wt.copy_(torch.randn(cout, cin, k, k, device=dev))
        pk = tconv.pack(wt, tconv.FWD)
        y = ops.empty_act(n, cout, h, w, dev)
        seg = tconv.Segment(x, k, (k - 1) // 2, True, 0)
        fl = 2.0 * n * h * w * cout * k * k * cin
        with torch.no_grad():
            us_old = timeit(lambda: ops.Conv2dFn.apply(x, wt, None, 1, (k - 1) // 2, L.PAD_REFLECT, L.ACT_RELU, 0.0))
            us_new = timeit(lambda: tconv.run([seg], pk, None, y, cout, n, h, w, h, w, L.ACT_RELU))
            us_pack = timeit(lambda: tconv.pack_into(pk, wt, tconv.FWD))
        print(f'{name:16s} im2col {us_old:8.1f} us {fl / us_old / 1e6:7.2f} TF | tconv {us_new:8.1f} us {fl / us_new / 1e6:7.2f} TF | pack {us_pack:6.1f} us'
              f'  (CAT_PK_TW={os.environ.get("CAT_PK_TW", "auto")})')
    # the student block's branch sum: 6 second convs K-concatenated vs 6 convs + add_n
    ms, kss, cout = [11, 12, 18, 15, 15, 12], [1, 3, 5, 1, 1, 1], 77
    offs = np.cumsum([0] + [tconv.cs4(m) for m in ms])
    hc = int(offs[-1])
    hg = torch.randn(n, h, w, hc, device=dev)
    scg, shg = torch.rand(hc, device=dev) + 0.5, torch.randn(hc, device=dev) * 0.1
    packs, segs, poff, fl = [], [], 0, 0.0
    for bi, (m, k) in enumerate(zip(ms, kss)):
        wt = ops.padded_weight_like((cout, m, k, k), dev)
        wt.copy_(torch.randn(cout, m, k, k, device=dev))
        packs.append(tconv.pack(wt, tconv.FWD))
        o = int(offs[bi])
        segs.append(tconv.Segment(hg, k, (k - 1) // 2, True, poff, c4=tconv.cs4(m), scale=scg[o:], shift=shg[o:], act=L.ACT_RELU, xcs=hc,
                                  ptr=hg.data_ptr() + 4 * o))
        poff += packs[-1].numel()
        fl += 2.0 * n * h * w * cout * k * k * m
    pk = torch.cat(packs)
    y = ops.empty_act(n, cout, h, w, dev)
    us = timeit(lambda: tconv.run(segs, pk, None, y, cout, n, h, w, h, w))
    print(f'S block branch sum (6 segments, K = {sum(m * k * k for m, k in zip(ms, kss))}) {us:8.1f} us {fl / us / 1e6:7.2f} TF')


if __name__ == '__main__':
    L.load()
    sys.exit(bench() if '--bench' in sys.argv else check())
