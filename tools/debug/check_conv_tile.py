"""GPU check of the opt-in LDS-tile conv kernel (conv_tile.hip) against ATen on the host; run with CAT_CONV_TILE=2 (force).
Exits non-zero on a mismatch or if the tile kernel was not the one that ran.  Also prints its time next to the im2col kernel's
on the layers it is meant for when called with --bench (two processes: CAT_CONV_TILE=0 / 1)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cat_amd import _lib as L, ops, synthetic  # noqa: E402

CASES = [  # cin, cout, k, reflect, act, n, h, w
    (54, 7, 5, True, 0, 2, 16, 16), (96, 16, 3, True, 1, 1, 6, 6), (22, 18, 5, True, 0, 1, 9, 35), (10, 7, 3, False, 2, 2, 8, 32),
    (36, 42, 5, False, 1, 1, 11, 40), (16, 16, 3, True, 0, 1, 16, 33), (82, 17, 5, True, 1, 2, 24, 40), (256, 42, 3, True, 1, 1, 16, 64),
    (4, 1, 3, False, 0, 1, 5, 5), (77, 33, 5, False, 3, 1, 8, 8),
]


DGRAD_CASES = [  # cin, cout, k, reflect, n, h, w   (forward of these runs on the im2col kernel: Cout > 48)
    (18, 82, 5, True, 1, 9, 33), (12, 54, 3, False, 2, 8, 20), (17, 50, 5, False, 1, 10, 34), (32, 56, 3, True, 1, 16, 40),
]


WGRAD_CASES = [  # cin, cout, k, reflect, n, h, w
    (22, 18, 5, True, 1, 9, 35), (18, 40, 3, False, 2, 8, 40), (6, 7, 5, False, 1, 16, 32), (82, 17, 5, True, 2, 16, 40), (17, 82, 3, True, 1, 8, 64),
]


def families():
    lib = L.load()
    n = lib.cat_prof_collect()
    name, cnt, ms, fl = C.create_string_buffer(64), C.c_int64(), C.c_double(), C.c_double()
    out = {}
    for i in range(n):
        lib.cat_prof_family(i, name, 64, C.byref(cnt), C.byref(ms), C.byref(fl))
        out[name.value.decode()] = (cnt.value, ms.value, fl.value)
    return out


def main():
    lib = L.load()
    dev = torch.device('cuda:0')
    if '--bench' in sys.argv:
        shapes = [('S 82->17 k5 @64', 82, 17, 5, 16, 64, 64), ('S 82->17 k3 @64', 82, 17, 3, 16, 64, 64), ('S 82->14 k5 @64', 82, 14, 5, 16, 64, 64),
                  ('T 256->42 k5 @64', 256, 42, 5, 16, 64, 64), ('T 256->42 k3 @64', 256, 42, 3, 16, 64, 64)]
        for name, cin, cout, k, n, h, w in shapes:
            x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
            wt = ops.padded_weight_like((cout, cin, k, k), dev)
            wt.copy_(torch.randn(cout, cin, k, k, device=dev))
            fn = lambda: ops.Conv2dFn.apply(x, wt, None, 1, (k - 1) // 2, L.PAD_REFLECT, L.ACT_RELU, 0.0)
            with torch.no_grad():
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            print(f'{name:20s} {us:9.1f} us  {2.0 * n * h * w * cout * k * k * cin / us / 1e6:7.2f} TFLOP/s  (CAT_CONV_TILE={os.environ.get("CAT_CONV_TILE", "0")})')
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for name, cin, cout, k, n, h, w in [('S wgrad 82->17 k5 @64', 82, 17, 5, 16, 64, 64), ('S wgrad 82->17 k3 @64', 82, 17, 3, 16, 64, 64),
                                            ('S wgrad 17->82 k5 @64', 17, 82, 5, 16, 64, 64), ('S wgrad 17->82 k3 @64', 17, 82, 3, 16, 64, 64)]:
            p = (k - 1) // 2
            x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
            dy = ops.to_nhwc(torch.randn(n, cout, h, w, device=dev))
            dw = ops.padded_weight_like((cout, cin, k, k), dev)
            g = L.ConvGeom(n, h, w, cin, ops.act_cs(x), h, w, cout, ops.act_cs(dy), k, k, 1, p, L.PAD_REFLECT, 0, 0.0, ops.act_cs(dy), ops.weight_wcs(dw))
            ws = torch.empty(max(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)) // 4, 1), device=dev)
            P = lambda t: C.c_void_p(t.data_ptr())
            fn = lambda: L.call('cat_conv2d_wgrad', C.byref(g), P(x), P(dy), P(dw), 0, P(ws), st)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            print(f'{name:20s} {us:9.1f} us  {2.0 * n * h * w * cout * k * k * cin / us / 1e6:7.2f} TFLOP/s  (CAT_CONV_TILE={os.environ.get("CAT_CONV_TILE", "0")})')
        for name, cin, cout, k, n, h, w in [('S dgrad 17<-82 k5 @64', 17, 82, 5, 16, 64, 64), ('S dgrad 17<-82 k3 @64', 17, 82, 3, 16, 64, 64)]:
            p = (k - 1) // 2
            dy = ops.to_nhwc(torch.randn(n, cout, h, w, device=dev))
            wt = ops.padded_weight_like((cout, cin, k, k), dev)
            wt.copy_(torch.randn(cout, cin, k, k, device=dev))
            dx = ops.empty_act(n, cin, h + 2 * p, w + 2 * p, dev)
            g = L.ConvGeom(n, h, w, cin, ops.cs_for(cin), h, w, cout, ops.act_cs(dy), k, k, 1, p, L.PAD_REFLECT, 0, 0.0, ops.act_cs(dy), ops.weight_wcs(wt))
            fn = lambda: ops._conv_dgrad(g, dy, wt, None, dx, ops.act_cs(dx), ops.act_cs(dx), st)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            print(f'{name:20s} {us:9.1f} us  {2.0 * n * h * w * cout * k * k * cin / us / 1e6:7.2f} TFLOP/s  (CAT_CONV_TILE={os.environ.get("CAT_CONV_TILE", "0")})')
        return 0
    lib.cat_prof_enable(1)
    worst = 0.0
    for cin, cout, k, reflect, act, n, h, w in CASES:
        pad = (k - 1) // 2
        x = synthetic.normal((n, cin, h, w), 1)
        wt = synthetic.normal((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
        b = synthetic.normal((cout,), 3, 0.1)
        xp = F.pad(x, (pad,) * 4, mode='reflect') if reflect else x
        yr = F.conv2d(xp, wt, b, padding=0 if reflect else pad)
        yr = {0: yr, 1: F.relu(yr), 2: F.leaky_relu(yr, 0.2), 3: torch.tanh(yr)}[act]
        wg = ops.padded_weight_like((cout, cin, k, k), dev)      # the product's parameter layout: [O][kh][kw][round_up(I, 4)]
        wg.copy_(wt)
        with torch.no_grad():
            y = ops.Conv2dFn.apply(ops.to_nhwc(x.to(dev)), wg, b.to(dev), 1, pad, 1 if reflect else 0, act, 0.2)
        err = float((y.cpu() - yr).abs().max() / yr.abs().max())
        cs = ops.act_cs(y)
        full = torch.as_strided(y, (y.shape[0], cs, y.shape[2], y.shape[3]), y.stride())
        padz = float(full[:, cout:].abs().max()) if cs > cout else 0.0
        print(f'{cin}->{cout} k{k} reflect={reflect} act={act} {n}x{h}x{w}: rel err {err:.2e}, pad lanes {padz}')
        worst = max(worst, err, padz)
    # dgrad convention: gradient w.r.t. the input of convs with few INPUT channels (GEMM-N = Cin <= 32)
    for cin, cout, k, reflect, n, h, w in DGRAD_CASES:
        pad = (k - 1) // 2
        x = synthetic.normal((n, cin, h, w), 5)
        wt = synthetic.normal((cout, cin, k, k), 6, 1.0 / np.sqrt(cin * k * k))
        gy = synthetic.normal((n, cout, h, w), 7)
        xr = x.clone().requires_grad_(True)
        xp = F.pad(xr, (pad,) * 4, mode='reflect') if reflect else xr
        F.conv2d(xp, wt, None, padding=0 if reflect else pad).backward(gy)
        wg = ops.padded_weight_like((cout, cin, k, k), dev)
        wg.copy_(wt)
        xg = ops.to_nhwc(x.to(dev)).detach().requires_grad_(True)
        y = ops.Conv2dFn.apply(xg, wg, None, 1, pad, 1 if reflect else 0, 0, 0.0)
        y.backward(ops.to_nhwc(gy.to(dev)))
        err = float((xg.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max())
        print(f'dgrad {cin}<-{cout} k{k} reflect={reflect} {n}x{h}x{w}: rel err {err:.2e}')
        worst = max(worst, err)
    # wgrad from LDS tiles: layers that are narrow on one side
    for cin, cout, k, reflect, n, h, w in WGRAD_CASES:
        pad = (k - 1) // 2
        x = synthetic.normal((n, cin, h, w), 8)
        wt = synthetic.normal((cout, cin, k, k), 9, 1.0 / np.sqrt(cin * k * k))
        gy = synthetic.normal((n, cout, h, w), 10)
        wr = wt.clone().requires_grad_(True)
        xp = F.pad(x, (pad,) * 4, mode='reflect') if reflect else x
        F.conv2d(xp, wr, None, padding=0 if reflect else pad).backward(gy)
        wg = ops.padded_weight_like((cout, cin, k, k), dev)
        wg.copy_(wt)
        wg.requires_grad_(True)
        y = ops.Conv2dFn.apply(ops.to_nhwc(x.to(dev)), wg, None, 1, pad, 1 if reflect else 0, 0, 0.0)
        y.backward(ops.to_nhwc(gy.to(dev)))
        err = float((wg.grad.cpu() - wr.grad).abs().max() / wr.grad.abs().max())
        print(f'wgrad {cin}->{cout} k{k} reflect={reflect} {n}x{h}x{w}: rel err {err:.2e}')
        worst = max(worst, err)
    torch.cuda.synchronize()
    fam = families()
    lib.cat_prof_enable(0)
    used = fam.get('conv_fwd_tile', (0, 0, 0))[0]
    used_d = fam.get('conv_dgrad_tile', (0, 0, 0))[0]
    used_w = fam.get('conv_wgrad_tile', (0, 0, 0))[0]
    print('conv_fwd_tile launches:', used, 'conv_dgrad_tile launches:', used_d, 'conv_wgrad_tile launches:', used_w, '| other conv families:',
          sorted(k for k in fam if k.startswith('conv_') and k not in ('conv_fwd_tile', 'conv_dgrad_tile', 'conv_wgrad_tile')))
    n_fwd = len(CASES) + sum(1 for c in WGRAD_CASES if c[1] <= 48)      # the wgrad cases' forward passes may use the tile kernel too
    ok = worst < 1e-4 and used == n_fwd and used_d == len(DGRAD_CASES) and used_w == len(WGRAD_CASES)
    print('OK' if ok else 'FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
