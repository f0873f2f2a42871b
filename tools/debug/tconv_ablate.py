"""Where does tconv_kernel's time go?  Times representative launches under the kernel's diagnostic switches (CAT_PK_ABLATE bits, results
become wrong, the instruction stream stays the same): 1 = every MFMA group reads the SAME 1 KB filter block per N tile (filter stream from
L1), 2 = every A fragment reads LDS offset 0 (no bank conflicts), 4 = staging loads hit one cached line (no HBM / L2 patch traffic),
8 = no staging / barrier after the first chunk.      python tools/debug/tconv_ablate.py"""
import os
import sys

os.environ.setdefault('CAT_LIB', 'diag')      # the ablation switches exist only in the diagnostic build: python -m cat_amd._build --diag

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cat_amd import _lib as L, ops, tconv  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=30):
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
    L.load()
    n, h, w = 16, 64, 64
    cases = []
    for name, cin, cout, k in [('T 256->42 k5 (NT3)', 256, 42, 5), ('T 42->256 k5 (NT8)', 42, 256, 5), ('T 256->42 k3 (NT3)', 256, 42, 3),
                               ('S 77->18 k5 (NT2)', 77, 18, 5), ('S 18->77 k5 (NT5)', 18, 77, 5)]:
        x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
        wt = ops.padded_weight_like((cout, cin, k, k), dev)
        wt.copy_(torch.randn(cout, cin, k, k, device=dev))
        pk = tconv.pack(wt, tconv.FWD)
        y = ops.empty_act(n, cout, h, w, dev)
        seg = tconv.Segment(x, k, (k - 1) // 2, True, 0)
        fl = 2.0 * n * h * w * cout * k * k * cin
        cases.append((name, fl, (lambda seg=seg, pk=pk, y=y, cout=cout: tconv.run([seg], pk, None, y, cout, n, h, w, h, w, L.ACT_RELU))))
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
    cases.append(('S branch sum (NT5, 6 seg)', fl, lambda: tconv.run(segs, pk, None, y, cout, n, h, w, h, w)))
    # CAT_PK_ABLATE is read once per process (static in the library): one child process per mode
    mode = os.environ.get('CAT_PK_ABLATE_CHILD')
    if mode is not None:
        with torch.no_grad():
            print('ROW ' + ' '.join('%.2f' % timeit(fn) for _, _, fn in cases))
        return
    import subprocess
    modes = [0, 1, 2, 4, 8, 3, 12, 15]
    cols = []
    for m in modes:
        env = dict(os.environ, CAT_PK_ABLATE=str(m), CAT_PK_ABLATE_CHILD='1')
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True).stdout
        cols.append([float(v) for v in next(ln for ln in out.splitlines() if ln.startswith('ROW ')).split()[1:]])
    print('%-28s' % 'launch' + ''.join('%11s' % ('abl=%d' % m) for m in modes) + '    (us; TFLOP/s at abl=0)')
    for i, (name, fl, _) in enumerate(cases):
        row = [c[i] for c in cols]
        print('%-28s' % name + ''.join('%11.1f' % t for t in row) + '    %.1f TF' % (fl / row[0] / 1e6))


if __name__ == '__main__':
    main()
