"""Time the wide-tile K-concatenated sum (cat_conv2d_ksum_fwd) on the teacher's block tail and related shapes; optionally next to the
LDS-tile launch it replaces.      python tools/debug/ksum_bench.py [--tconv] [--iters 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cat_amd import _lib as L, ops, ksum, tconv

dev = torch.device('cuda:0')


def timeit(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--tconv', action='store_true')
    args = ap.parse_args()
    L.load()
    cases = [('T tail 176x1 + 42x3 + 42x5 -> 256', [(176, 1), (42, 3), (42, 5)], 256, 16, 64, 64),
             ('42x5 -> 256', [(42, 5)], 256, 16, 64, 64),
             ('42x3 -> 256', [(42, 3)], 256, 16, 64, 64),
             ('176x1 -> 256', [(176, 1)], 256, 16, 64, 64),
             ('256x3 -> 256 (32-ch chunks only)', [(256, 3)], 256, 16, 64, 64),
             ('T tail @ batch 8', [(176, 1), (42, 3), (42, 5)], 256, 8, 64, 64)]
    for name, segs, cout, n, h, w in cases:
        ks, fl = [], 0.0
        for c, k in segs:
            wd = ops.padded_weight_like((cout, c, k, k), dev)
            wd.copy_(torch.randn(cout, c, k, k, device=dev) / (c * k * k) ** 0.5)
            ks.append(ksum.Segment(ops.to_nhwc(torch.randn(n, c, h, w, device=dev)), wd, True))
            fl += 2.0 * n * h * w * cout * k * k * c
        y = ops.empty_act(n, cout, h, w, dev)
        res = ops.to_nhwc(torch.randn(n, cout, h, w, device=dev))
        bias = torch.randn(cout, device=dev)
        us = timeit(lambda: ksum.run(ks, bias, y, res=res), args.iters)
        line = '%-40s ksum %8.1f us %7.1f TF' % (name, us, fl / us / 1e6)
        if args.tconv:
            packs = [tconv.pack(s.w, tconv.FWD) for s in ks]
            offs, po = [], 0
            for pk in packs:
                offs.append(po)
                po += pk.numel()
            pack = torch.cat(packs)
            tsegs = [tconv.Segment(s.src, s.ks, (s.ks - 1) // 2, s.ks > 1, off) for s, off in zip(ks, offs)]
            us2 = timeit(lambda: tconv.run(tsegs, pack, bias, y, cout, n, h, w, h, w, res=res), args.iters)
            line += '   tconv %8.1f us %7.1f TF' % (us2, fl / us2 / 1e6)
        print(line)


if __name__ == '__main__':
    main()
