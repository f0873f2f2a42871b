"""Host side of the quad-granule LDS-tile convolution (csrc/conv_q.hip, cat_qconv_* in include/cat_hip.h).

A launch = K segments (source, tap rectangle, optional staging affine + activation) summed into one output lattice, with a source stride
(the generator's stride-2 3x3 convs) or four sub-pixel output classes (its ConvTranspose2d layers), reference
models/modules/inception_architecture/inception_generator.py:37-56,118-132.  The filter stream's layout follows the launch plan
(cat_qconv_plan), so a layer is described once (`Layer`), planned once per input geometry and re-packed when its weight changes."""
import ctypes as C

import torch

from . import _lib as L
from . import ops


def cs4(c):
    return (c + 3) // 4 * 4


class Seg:
    """One K segment: tap (i, j) of lattice pixel (cy, cx) reads src pixel (cy*S + oy + i, cx*S + ox + j)."""
    __slots__ = ('ptr', 'xcs', 'c4', 'cin', 'kh', 'kw', 'oy', 'ox', 'reflect', 'scale', 'shift', 'sstride', 'act', 'slope', 'pack_off')

    def __init__(self, src, kh, kw, oy, ox, reflect=False, scale=None, shift=None, sstride=0, act=L.ACT_NONE, slope=0.0, c4=None, cin=None,
                 xcs=None, ptr=None, pack_off=0):
        self.ptr = ptr if ptr is not None else src.data_ptr()
        self.xcs = xcs if xcs is not None else ops.act_cs(src)
        self.cin = cin if cin is not None else src.shape[1]
        self.c4 = c4 if c4 is not None else cs4(self.cin)
        self.kh, self.kw, self.oy, self.ox, self.reflect = kh, kw, oy, ox, int(reflect)
        self.scale, self.shift, self.sstride, self.act, self.slope, self.pack_off = scale, shift, sstride, act, slope, int(pack_off)


def _ptr(t):
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


def geometry(segs, n, h, w, ho, wo, nn, ycs, ycw=None, stride=1, ncls=1, act=L.ACT_NONE, slope=0.0, res=None, stats=None, scs=0, nvalid=0):
    g = L.QConv()
    g.N, g.H, g.W, g.Ho, g.Wo, g.S = n, h, w, ho, wo, stride
    g.OS, g.ncls = (2, 4) if ncls == 4 else (1, 1)
    g.Nn, g.ycs, g.ycw, g.nvalid, g.act, g.slope, g.nseg = nn, ycs, ycw if ycw is not None else cs4(nn), nvalid, act, slope, len(segs)
    if res is not None:
        g.res, g.rcs = res.data_ptr(), ops.act_cs(res)
    if stats is not None:
        g.stats, g.scs = _ptr(stats), scs
    if len(segs) > L.QCONV_MAXSEG:
        raise RuntimeError('qconv: too many K segments')
    for k, s in enumerate(segs):
        t = g.seg[k]
        t.src, t.scale, t.shift = s.ptr, _ptr(s.scale), _ptr(s.shift)
        t.sstride, t.xcs, t.c4, t.cin, t.kh, t.kw, t.oy, t.ox = s.sstride, s.xcs, s.c4, s.cin, s.kh, s.kw, s.oy, s.ox
        t.act, t.slope, t.reflect, t.pack_off = s.act, s.slope, s.reflect, s.pack_off
    return g


def _tile_rule():
    """The library's mutable tile rule (cat_qconv_min_tiles16: from how many 16-wide tiles a layer takes 4-row pixel groups).  A plan -- tile
    height, quads per wave, layout of the packed stream, size of the statistics table -- depends on it, and cat_qconv_fwd re-plans from the
    CURRENT value: cached plans and packed streams are keyed on it, so a runtime change never pairs a stale plan with a new launch."""
    return int(L.query('cat_qconv_min_tiles16', -1))


def plan_of(g):
    p = L.QPlan()
    L.call('cat_qconv_plan', C.byref(g), C.byref(p))
    return p


def launch(g, pack, bias, y_ptr):
    L.call('cat_qconv_fwd', C.byref(g), ops._p(pack), None if bias is None else C.c_void_p(_ptr(bias)), C.c_void_p(y_ptr), ops._stream())


def pack_conv(g, seg, dst, wcl, wcs, nn, ntaps):
    """nn.Conv2d weight [Nn][kh*kw][wcs] -> segment `seg`'s part of the launch's stream."""
    L.call('cat_qconv_pack', C.byref(g), seg, ops._p(wcl), C.c_void_p(_ptr(dst)), nn, None, ntaps * wcs, wcs, 1, ops._stream())


# sub-pixel classes of ConvTranspose2d(k 3, stride 2, padding 1, output_padding 1): output row 2a + py reads input rows a (+1) with
# filter rows ky:  py = 0 -> (dy 0, ky 1);  py = 1 -> (dy 0, ky 2), (dy 1, ky 0).  Same along x.
_CT_K = {0: [1], 1: [2, 0]}


def ct_class_taps(py, px):
    """filter tap index ky*3 + kx for every tap (i, j) of class (py, px)'s (1 + py) x (1 + px) rectangle"""
    return [ky * 3 + kx for ky in _CT_K[py] for kx in _CT_K[px]]


def pack_conv_transpose(g, dst, wcl, wcs, nn):
    """nn.ConvTranspose2d weight [Cin][3*3][wcs >= Nn]: segment 2*py+px = sub-pixel class (py, px)."""
    for cls in range(4):
        taps = ct_class_taps(cls >> 1, cls & 1)
        arr = (C.c_int * len(taps))(*taps)
        L.call('cat_qconv_pack', C.byref(g), cls, ops._p(wcl), C.c_void_p(_ptr(dst)), nn, arr, 1, wcs, 9 * wcs, ops._stream())


class Layer:
    """A Conv2d / ConvTranspose2d of the generator's edge on the quad-granule kernel: geometry + plan per input shape, packed filters per
    weight version.  kind 'conv' (any kernel size <= 7, stride 1 | 2, zero or reflect padding) or 'convt' (k 3, s 2, p 1, op 1)."""

    def __init__(self, kind, weight, stride=1, pad=0, reflect=False):
        self.kind, self.weight, self.stride, self.pad, self.reflect = kind, weight, stride, pad, reflect
        self._plans = {}
        self._pk = None      # (key, tensor, plan signature)

    @staticmethod
    def supported(kind, weight, stride, pad, out_pad=0):
        if kind == 'conv':
            o, i, kh, kw = weight.shape
            return kh == kw and kh in (1, 3, 5, 7) and stride in (1, 2) and pad == (kh - 1) // 2 and (stride == 1 or kh == 3)
        i, o, kh, kw = weight.shape
        return kh == kw == 3 and stride == 2 and pad == 1 and out_pad == 1

    def out_shape(self, x):
        n, c, h, w = x.shape
        if self.kind == 'conv':
            k = self.weight.shape[2]
            return self.weight.shape[0], (h + 2 * self.pad - k) // self.stride + 1, (w + 2 * self.pad - k) // self.stride + 1
        return self.weight.shape[1], 2 * h, 2 * w

    def _segments(self, x, pre):
        """pre = (scale, shift, sstride, act, slope) of a pending norm + activation on x, or None"""
        kw = {}
        if pre is not None:
            kw = dict(scale=pre[0], shift=pre[1], sstride=pre[2], act=pre[3], slope=pre[4])
        if self.kind == 'conv':
            k = self.weight.shape[2]
            return [Seg(x, k, k, -self.pad, -self.pad, reflect=self.reflect, **kw)]
        return [Seg(x, 1 + (c >> 1), 1 + (c & 1), 0, 0, **kw) for c in range(4)]

    def plan_for(self, x):
        """The launch plan for an input of x's shape (tile geometry / entries of the statistics table) without launching."""
        n, c, h, w = x.shape
        key = (n, h, w, c, _tile_rule())
        plan = self._plans.get(key)
        if plan is None:
            nn, ho, wo = self.out_shape(x)
            ct = self.kind == 'convt'
            g = geometry(self._segments(x, None), n, h, w, h if ct else ho, w if ct else wo, nn, cs4(nn), stride=1 if ct else self.stride,
                         ncls=4 if ct else 1)
            plan = self._plans[key] = plan_of(g)
        return plan

    def run(self, x, bias, y, act=L.ACT_NONE, slope=0.0, pre=None, stats=None, scs=0):
        """Returns the plan (tile geometry of the statistics table).  y: NHWC output activation."""
        from . import optim
        n, c, h, w = x.shape
        nn, ho, wo = self.out_shape(x)
        segs = self._segments(x, pre)
        ct = self.kind == 'convt'
        lat_h, lat_w = (h, w) if ct else (ho, wo)
        g = geometry(segs, n, h, w, lat_h, lat_w, nn, ops.act_cs(y), stride=1 if ct else self.stride, ncls=4 if ct else 1, act=act, slope=slope,
                     stats=stats, scs=scs)
        key = (n, h, w, c, _tile_rule())
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = plan_of(g)
        # packed stream: keyed like ops.packed_filter (version counter, or the optimizer epoch for FusedAdam-owned parameters)
        wcl, wcs = ops.weight_cl(self.weight)
        trainable = getattr(self.weight, '_cat_grad_view', None) is not None
        sig = (plan.cs, plan.nq, plan.nsplit, int(plan.pack_floats), plan.th)
        wkey = (wcl.data_ptr(), wcl._version, optim.weights_epoch() if trainable else -1, sig, key)
        if self._pk is None or self._pk[0] != wkey:
            same_layout = self._pk is not None and self._pk[0][3:] == wkey[3:] and self._pk[1].device == x.device
            # a new weight version re-packs in place; another geometry gets a fresh zeroed stream (its spare table rows / filter step must read 0)
            buf = self._pk[1] if same_layout else torch.zeros(int(plan.pack_floats), device=x.device, dtype=torch.float32)
            if ct:
                pack_conv_transpose(g, buf, wcl, wcs, nn)
            else:
                k = self.weight.shape[2]
                pack_conv(g, 0, buf, wcl, wcs, nn, k * k)
            self._pk = (wkey, buf)
        launch(g, self._pk[1], bias, y.data_ptr())
        return plan
