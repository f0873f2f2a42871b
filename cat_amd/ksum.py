"""Host side of the K-concatenated wide-tile convolution sum (csrc/conv_ksum.hip, cat_conv2d_ksum_* in include/cat_hip.h):

    y = act(bias + sum_s conv_{k_s x k_s, stride 1, "same"}(src_s, w_s)) + res

as ONE implicit GEMM on 128 x 128 tiles -- the tail of the frozen (eval-mode, BatchNorm-folded) teacher's InvertedResidualChannels block
(reference models/modules/inception_modules.py:230-236; cat_amd/frozen.py)."""
import ctypes as C
import os

from . import _lib as L
from . import ops

_ENABLED = os.environ.get('CAT_KSUM', '1') != '0'      # A/B switch: 0 keeps the LDS-tile launch (cat_tconv_fwd)
_MIN_WG = int(os.environ.get('CAT_KSUM_MIN_WG', '256'))


def set_min_workgroups(n):
    """Fewest 128 x 128 output tiles for which the wide-tile sum is chosen (tests lower it to drive small planes through the kernel)."""
    global _MIN_WG
    old, _MIN_WG = _MIN_WG, int(n)
    return old


class Segment:
    """src: NHWC activation [n, c, h, w] (or a channel slice of one); w: conv weight [Cout, c, k, k] in kernel layout."""
    __slots__ = ('src', 'w', 'ks', 'reflect')

    def __init__(self, src, w, reflect):
        self.src, self.w, self.ks, self.reflect = src, w, int(w.shape[2]), int(bool(reflect))


def _geom(segs, n, h, w, cout, ycs, ycw, rcs, act, slope):
    g = L.KSum()
    g.N, g.H, g.W, g.Cout, g.ycs, g.ycw, g.rcs, g.act, g.slope, g.nseg = n, h, w, cout, ycs, ycw, rcs, act, slope, len(segs)
    keep = []
    for k, s in enumerate(segs):
        wcl, wcs = ops.weight_cl(s.w)
        keep.append(wcl)
        t = g.seg[k]
        cin = s.src.shape[1]
        t.src, t.w = s.src.data_ptr(), wcl.data_ptr()
        t.xcs, t.c4, t.cin, t.ks, t.reflect, t.wcs = ops.act_cs(s.src), ops.cs_for(cin), cin, s.ks, s.reflect, wcs
    return g, keep


def applicable(segs, n, h, w, cout):
    if not _ENABLED or len(segs) > L.KSUM_MAXSEG or cout < 64:
        return False
    if ((n * h * w + 127) // 128) * ((cout + 127) // 128) < _MIN_WG:
        return False
    for s in segs:
        if s.w.shape[2] != s.w.shape[3] or s.w.shape[0] != cout or s.w.shape[1] != s.src.shape[1] or not ops.is_act(s.src):
            return False
        if ops.weight_wcs(s.w) is None or ops.weight_wcs(s.w) % 4:
            return False
    g, _ = _geom(segs, n, h, w, cout, ops.cs_for(cout), ops.cs_for(cout), 0, L.ACT_NONE, 0.0)
    return bool(L.query('cat_conv2d_ksum_supported', C.byref(g)))


def run(segs, bias, y, res=None, act=L.ACT_NONE, slope=0.0):
    """Enqueue the launch.  y: NHWC activation [n, cout, h, w]; res: optional residual of the same shape (added after the activation)."""
    n, cout, h, w = y.shape
    g, keep = _geom(segs, n, h, w, cout, ops.act_cs(y), ops.act_cs(y), 0 if res is None else ops.act_cs(res), act, slope)
    L.call('cat_conv2d_ksum_fwd', C.byref(g), None if bias is None else ops._p(bias), None if res is None else ops._p(res), ops._p(y), ops._stream())
    return y
