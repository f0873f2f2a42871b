"""Per-layer time of the quad-granule kernel (csrc/conv_q.hip) next to the kernel the layer ran on before (cat_amd.ops dispatch: im2col
implicit GEMM, LDS-tile tconv, smallco) -- each kernel timed ALONE with HIP events, batch 16 at the bench's planes.
    python tools/qconv_bench.py [--match stem] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cat_amd import _lib as L, ops, qconv  # noqa: E402

# name, kind, N, H, W, Cin, Cout, k, stride, reflect, act
LAYERS = [
    ('S.stem 3->25 k7 @256', 'conv', 16, 256, 256, 3, 25, 7, 1, 1, 0),
    ('S.down1 25->40 k3 s2 @256', 'conv', 16, 256, 256, 25, 40, 3, 2, 0, 0),
    ('S.down2 40->77 k3 s2 @128', 'conv', 16, 128, 128, 40, 77, 3, 2, 0, 0),
    ('S.up1 77->38 ct @64', 'convt', 16, 64, 64, 77, 38, 3, 2, 0, 0),
    ('S.up2 38->23 ct @128', 'convt', 16, 128, 128, 38, 23, 3, 2, 0, 0),
    ('S.head 23->3 k7 @256', 'conv', 16, 256, 256, 23, 3, 7, 1, 1, 3),
    ('S.res5a 77->18 k5 @64', 'conv', 16, 64, 64, 77, 18, 5, 1, 1, 0),
    ('S.res5b 18->77 k5 @64', 'conv', 16, 64, 64, 18, 77, 5, 1, 1, 0),
    ('S.res3a 77->12 k3 @64', 'conv', 16, 64, 64, 77, 12, 3, 1, 1, 0),
    ('S.pw 77->54 k1 @64', 'conv', 16, 64, 64, 77, 54, 1, 1, 0, 0),
    ('T.res5a 256->42 k5 @64', 'conv', 16, 64, 64, 256, 42, 5, 1, 1, 0),
    ('T.res5b 42->256 k5 @64', 'conv', 16, 64, 64, 42, 256, 5, 1, 1, 0),
    ('T.res3a 256->42 k3 @64', 'conv', 16, 64, 64, 256, 42, 3, 1, 1, 0),
    ('T.pw 256->176 k1 @64', 'conv', 16, 64, 64, 256, 176, 1, 1, 0, 0),
    ('T.stem 3->64 k7 @256', 'conv', 16, 256, 256, 3, 64, 7, 1, 1, 0),
    ('T.head 64->3 k7 @256', 'conv', 16, 256, 256, 64, 3, 7, 1, 1, 3),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--match', default='')
    args = ap.parse_args()
    L.load()
    dev = torch.device('cuda:0')
    print(f'{"layer":30s} {"old us":>8s} {"TF":>6s} | {"qconv us":>8s} {"TF":>6s} {"+stats":>8s} | plan (cs nq nsplit th) | max rel diff')
    for name, kind, n, h, w, cin, cout, k, stride, refl, act in LAYERS:
        if args.match and args.match not in name:
            continue
        pad = (k - 1) // 2
        x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
        if kind == 'conv':
            wt = ops.padded_weight_like((cout, cin, k, k), dev) if cin > 1 else torch.empty(cout, cin, k, k, device=dev)
            wt.copy_(torch.randn(cout, cin, k, k, device=dev) * (cin * k * k) ** -0.5)
            ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
            flops = 2.0 * n * ho * wo * cout * cin * k * k
            old = lambda: ops.Conv2dFn.apply(x, wt, None, stride, pad, L.PAD_REFLECT if refl else L.PAD_ZERO, act, 0.0)
        else:
            wt = ops.padded_weight_like((cin, cout, 3, 3), dev)
            wt.copy_(torch.randn(cin, cout, 3, 3, device=dev) * (cin * 9) ** -0.5)
            ho, wo = 2 * h, 2 * w
            flops = 2.0 * n * h * w * cout * cin * 9
            old = lambda: ops.ConvTranspose2dFn.apply(x, wt, None, 2, 1, 1)
        layer = qconv.Layer(kind, wt, stride=stride, pad=pad, reflect=bool(refl))
        y = ops.empty_act(n, cout, ho, wo, dev)
        scs = qconv.cs4(cout)
        stats = torch.empty(n * 8192 * 2 * scs // 4 + 1024, device=dev)
        with torch.no_grad():
            t_old = timeit(old, args.iters)
            yo = old()
            plan = layer.run(x, None, y, act=act)
            diff = float((y - yo).abs().max() / yo.abs().max())
            t_q = timeit(lambda: layer.run(x, None, y, act=act), args.iters)
            t_qs = timeit(lambda: layer.run(x, None, y, stats=stats, scs=scs), args.iters) if act == 0 else float('nan')
        print(f'{name:30s} {t_old:8.1f} {flops / t_old / 1e6:6.1f} | {t_q:8.1f} {flops / t_q / 1e6:6.1f} {t_qs:8.1f} | '
              f'{plan.cs:2d} {plan.nq:2d} {plan.nsplit:2d} {plan.th:2d} | {diff:.1e}', flush=True)


if __name__ == '__main__':
    main()
