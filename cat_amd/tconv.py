"""Host side of the LDS-tile convolution with pre-packed filters (csrc/conv_pk.hip, cat_tconv_* in include/cat_hip.h).

A launch is a list of K segments -- (source activation or channel slice, kernel size, optional staging affine + activation, packed
filters) -- accumulated into one output: a single stride-1 conv, its input gradient, or the K-concatenated branch sum of an
InvertedResidualChannels block (reference models/modules/inception_modules.py:230-236)."""
import ctypes as C

import torch

from . import _lib as L
from . import ops

FWD, DGRAD = 0, 1


def cs4(c):
    return (c + 3) // 4 * 4


def pack_floats(ks, c4, nn):
    return int(L.query('cat_tconv_pack_floats', ks, c4, nn))


def pack_into(dst, w, mode):
    """Pack conv weight `w` (logical [O, I, k, k], kernel layout [O][k][k][wcs]) into `dst` (flat fp32, pack_floats(...) long).
    mode FWD: output channels = O, reduction over I;  mode DGRAD: output channels = I, reduction over O (taps flipped)."""
    wcl, wcs = ops.weight_cl(w)
    o, i, kh, kw = w.shape
    if kh != kw:
        raise RuntimeError('tconv: square kernels only')
    nn, ck = (o, i) if mode == FWD else (i, o)
    L.call('cat_tconv_pack', ops._p(wcl), mode, nn, ck, kh, wcs, kh * kw * wcs, cs4(ck), ops._p(dst), ops._stream())
    return dst


def pack(w, mode):
    o, i, kh, kw = w.shape
    nn, ck = (o, i) if mode == FWD else (i, o)
    dst = torch.empty(pack_floats(kh, cs4(ck), nn), device=w.device, dtype=torch.float32)
    return pack_into(dst, w, mode)


class Segment:
    """One K segment.  `src`: NHWC activation (or a channel-slice view of one); `ptr_off`: extra float offset into src's storage
    (slice start), `c4`: channels read (multiple of 4)."""
    __slots__ = ('src', 'c4', 'cin', 'ks', 'padv', 'reflect', 'scale', 'shift', 'act', 'slope', 'pack_off', 'xcs', 'ptr', 'sstride')

    def __init__(self, src, ks, padv, reflect, pack_off, c4=None, scale=None, shift=None, act=L.ACT_NONE, slope=0.0, xcs=None, ptr=None,
                 cin=None, sstride=0):
        self.sstride = sstride
        self.src, self.ks, self.padv, self.reflect, self.pack_off = src, ks, padv, int(reflect), int(pack_off)
        self.c4 = c4 if c4 is not None else cs4(src.shape[1])
        self.cin = cin if cin is not None else (src.shape[1] if c4 is None else c4)
        self.scale, self.shift, self.act, self.slope = scale, shift, act, slope
        self.xcs = xcs if xcs is not None else ops.act_cs(src)
        self.ptr = ptr if ptr is not None else src.data_ptr()


def _ptr(t):
    return t if isinstance(t, int) else t.data_ptr()


def run(segs, packbuf, bias, y, nn, n, h, w, ho, wo, act=L.ACT_NONE, slope=0.0, ycs=None, ycw=None, yptr=None, res=None, stats=None, scs=0, nvalid=0):
    """Enqueue one tconv launch.  y: NHWC activation [n, nn, ho, wo] (or pass yptr / ycs / ycw for a slice of a wider buffer)."""
    g = L.TConv()
    g.N, g.H, g.W, g.Ho, g.Wo = n, h, w, ho, wo
    g.Nn = nn
    g.nvalid = nvalid
    g.ycs = ycs if ycs is not None else ops.act_cs(y)
    g.ycw = ycw if ycw is not None else g.ycs
    g.act, g.slope, g.nseg = act, slope, len(segs)
    if res is not None:
        g.res, g.rcs = res.data_ptr(), ops.act_cs(res)
    if stats is not None:
        g.stats, g.scs = _ptr(stats), scs
    if len(segs) > L.TCONV_MAXSEG:
        raise RuntimeError('tconv: too many K segments')
    for k, s in enumerate(segs):
        t = g.seg[k]
        t.src = s.ptr
        t.scale = None if s.scale is None else _ptr(s.scale)
        t.shift = None if s.shift is None else _ptr(s.shift)
        t.sstride = s.sstride
        t.xcs, t.c4, t.cin, t.ks, t.padv, t.act, t.slope, t.reflect, t.pack_off = s.xcs, s.c4, s.cin, s.ks, s.padv, s.act, s.slope, s.reflect, s.pack_off
    L.call('cat_tconv_fwd', C.byref(g), ops._p(packbuf), None if bias is None else C.c_void_p(_ptr(bias)),
           C.c_void_p(yptr if yptr is not None else y.data_ptr()), ops._stream())
    return y
