"""Per-shape throughput of the implicit-GEMM conv kernels (fwd / dgrad / wgrad) on the layers that dominate the
distillation step.  GPU only:  python tools/conv_bench.py [--only fwd] [--iters 20] [--shapes big]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cat_amd import _lib as L  # noqa: E402
from cat_amd import ops  # noqa: E402

# name, N, H, W, Cin, Cout, k, stride, pad, reflect
SHAPES = {
    'big': [
        ('D.conv4 512->1024 k4 s1 @32', 16, 32, 32, 512, 1024, 4, 1, 1, 0),
        ('D.conv3 256->512 k4 s2 @64', 16, 64, 64, 256, 512, 4, 2, 1, 0),
        ('D.conv2 128->256 k4 s2 @128', 16, 128, 128, 128, 256, 4, 2, 1, 0),
        ('D.conv1 6->128 k4 s2 @256', 16, 256, 256, 6, 128, 4, 2, 1, 0),
        ('T.res5a 256->42 k5 @64', 16, 64, 64, 256, 42, 5, 1, 2, 1),
        ('T.res5b 42->256 k5 @64', 16, 64, 64, 42, 256, 5, 1, 2, 1),
        ('T.res3a 256->42 k3 @64', 16, 64, 64, 256, 42, 3, 1, 1, 1),
        ('T.res3b 42->256 k3 @64', 16, 64, 64, 42, 256, 3, 1, 1, 1),
        ('T.pw 256->42 k1 @64', 16, 64, 64, 256, 42, 1, 1, 0, 0),
        ('T.down2 128->256 k3 s2 @128', 16, 128, 128, 128, 256, 3, 2, 1, 0),
        ('T.down1 64->128 k3 s2 @256', 16, 256, 256, 64, 128, 3, 2, 1, 0),
        ('T.stem 3->64 k7 @256', 16, 256, 256, 3, 64, 7, 1, 3, 1),
        ('T.out 64->3 k7 @256', 16, 256, 256, 64, 3, 7, 1, 3, 1),
        ('D.conv5 1024->1 k4 s1 @31', 16, 31, 31, 1024, 1, 4, 1, 1, 0),
    ],
    'student': [
        ('S.res5a 77->18 k5 @64', 16, 64, 64, 77, 18, 5, 1, 2, 1),
        ('S.res5b 18->77 k5 @64', 16, 64, 64, 18, 77, 5, 1, 2, 1),
        ('S.res3a 77->12 k3 @64', 16, 64, 64, 77, 12, 3, 1, 1, 1),
        ('S.pw 77->15 k1 @64', 16, 64, 64, 77, 15, 1, 1, 0, 0),
        ('S.pwb 15->77 k1 @64', 16, 64, 64, 15, 77, 1, 1, 0, 0),
        ('S.stem 3->25 k7 @256', 16, 256, 256, 3, 25, 7, 1, 3, 1),
        ('S.out 23->3 k7 @256', 16, 256, 256, 23, 3, 7, 1, 3, 1),
    ],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='fwd,dgrad,wgrad')
    ap.add_argument('--shapes', default='big,student')
    ap.add_argument('--match', default='')
    args = ap.parse_args()
    L.load()
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    kinds = args.only.split(',')
    print(f'{"layer":34s} {"kind":6s} {"us":>9s} {"TFLOP/s":>8s} {"%peak":>6s}')
    for grp in args.shapes.split(','):
        for name, n, h, w, cin, cout, k, s, p, refl in SHAPES[grp]:
            if args.match and args.match not in name:
                continue
            ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
            wt = ops.padded_weight_like((cout, cin, k, k), dev)      # the product's weight layout: [O][kh][kw][round_up(I,4)]
            wt.copy_(torch.randn(cout, cin, k, k, device=dev))
            wcs = ops.weight_wcs(wt)
            y = ops.empty_act(n, cout, ho, wo, dev)
            dy = ops.to_nhwc(torch.randn(n, cout, ho, wo, device=dev))
            hp, wp = (h + 2 * p, w + 2 * p) if refl else (h, w)
            dx = ops.empty_act(n, cin, hp, wp, dev)
            dw = ops.padded_weight_like((cout, cin, k, k), dev)
            g = L.ConvGeom(n, h, w, cin, ops.act_cs(x), ho, wo, cout, ops.act_cs(y), k, k, s, p, 1 if refl else 0, 0, 0.0, ops.act_cs(y), wcs)
            ws = torch.empty(max(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)) // 4, 1), device=dev)
            flops = 2.0 * n * ho * wo * cout * k * k * cin
            P = lambda t: C.c_void_p(t.data_ptr())
            fns = {
                'fwd': lambda: ops._conv_fwd(g, x, wt, None, y, st),
                'dgrad': lambda: L.call('cat_conv2d_dgrad', C.byref(g), P(dy), P(wt), None, P(dx), ops.act_cs(dx), ops.act_cs(dx), st),
                'wgrad': lambda: L.call('cat_conv2d_wgrad', C.byref(g), P(x), P(dy), P(dw), 0, P(ws), st),
            }
            for kind in kinds:
                fn = fns[kind]
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                tf = flops / us / 1e6
                print(f'{name:34s} {kind:6s} {us:9.1f} {tf:8.2f} {100 * tf / 157.3:6.1f}')


if __name__ == '__main__':
    main()
