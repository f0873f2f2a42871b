"""wgrad (+ partial reduce) time per layer against the workgroup target of the pixel split (CAT_WGRAD_BLOCKS).  GPU only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402

from cat_amd import _lib as L, ops  # noqa: E402
import conv_bench  # noqa: E402

L.load()
dev = torch.device('cuda:0')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = conv_bench.SHAPES['student'][:5] + [s for s in conv_bench.SHAPES['big'] if s[0].startswith(('D.conv4', 'D.conv2', 'T.down1'))]
targets = [(1024, 256, 256), (1024, 64, 1024), (2048, 64, 1024), (1024, 128, 512)]
rows = []
for name, n, h, w, cin, cout, k, s, p, refl in shapes:
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = ops.to_nhwc(torch.randn(n, cin, h, w, device=dev))
    dy = ops.to_nhwc(torch.randn(n, cout, ho, wo, device=dev))
    dw = ops.padded_weight_like((cout, cin, k, k), dev)
    g = L.ConvGeom(n, h, w, cin, ops.act_cs(x), ho, wo, cout, ops.act_cs(dy), k, k, s, p, 1 if refl else 0, 0, 0.0, ops.act_cs(dy), ops.weight_wcs(dw))
    P = lambda t: C.c_void_p(t.data_ptr())
    res, ref = [], None
    for t in targets:
        os.environ['CAT_WGRAD_BLOCKS'] = str(t[0])
        os.environ['CAT_WGRAD_MINCHUNK'] = str(t[1])
        os.environ['CAT_WGRAD_MAXSPLIT'] = str(t[2])
        ws = torch.empty(max(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)) // 4, 1), device=dev)
        fn = lambda: L.call('cat_conv2d_wgrad', C.byref(g), P(x), P(dy), P(dw), 0, P(ws), st)
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
        if ref is None:
            ref = dw.clone()
        else:
            assert float((dw - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), (name, t)
    rows.append((name, res))
print('%-34s' % 'layer' + ''.join('%14s' % ('%d/%d/%d' % t) for t in targets))
for name, res in rows:
    print('%-34s' % name + ''.join('%14.1f' % r for r in res))
