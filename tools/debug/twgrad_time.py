"""Time cat_conv2d_wgrad on the student's narrow layers (kernel + reduce), events around 50 calls."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cat_amd import _lib as L, ops
L.load()
dev = torch.device('cuda')
for (cin, cout, k) in ((77, 15, 5), (15, 77, 5), (77, 12, 3), (12, 77, 3)):
    n, h, w = 16, 64, 64
    x = torch.randn(n, h, w, ops.cs_for(cin), device=dev)
    dy = torch.randn(n, h, w, ops.cs_for(cout), device=dev)
    dw = ops.padded_weight_like((cout, cin, k, k), dev)
    g = ops._conv_geom(n, h, w, cin, ops.cs_for(cin), h, w, cout, ops.cs_for(cout), k, k, 1, (k - 1) // 2, L.PAD_REFLECT, wcs=ops.cs_for(cin))
    ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)), dev)
    st = ops._stream()
    for _ in range(3):
        L.call('cat_conv2d_wgrad', C.byref(g), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.call('cat_conv2d_wgrad', C.byref(g), ops._p(x), ops._p(dy), ops._p(dw), 0, ops._p(ws), st)
    e1.record(); torch.cuda.synchronize()
    gf = 2.0 * n * h * w * cout * k * k * cin / 1e9
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f'{cin}->{cout} k{k}: {us:7.1f} us  {gf / us * 1e3:6.1f} TF   CAT_TWGRAD={os.environ.get("CAT_TWGRAD", "1")}')
