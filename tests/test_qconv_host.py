"""CPU: the quad-granule convolution's launch plan is a pure host function of the geometry (cat_qconv_plan): tile shape, stream size and the
statistics-table entries for the generator's edge layers, and its error behaviour -- no GPU, no kernel launch."""
import ctypes as C

import pytest

from cat_amd import _lib as L


def geom(n, h, w, ho, wo, cin, cout, k, stride=1, ncls=1):
    g = L.QConv()
    g.N, g.H, g.W, g.Ho, g.Wo, g.S = n, h, w, ho, wo, stride
    g.OS, g.ncls = (2, 4) if ncls == 4 else (1, 1)
    cs = (cout + 3) // 4 * 4
    g.Nn, g.ycs, g.ycw = cout, cs, cs
    c4 = (cin + 3) // 4 * 4
    if ncls == 4:
        g.nseg = 4
        for c in range(4):
            s = g.seg[c]
            s.kh, s.kw, s.oy, s.ox, s.c4, s.cin, s.xcs = 1 + (c >> 1), 1 + (c & 1), 0, 0, c4, cin, c4
    else:
        g.nseg = 1
        s = g.seg[0]
        s.kh, s.kw, s.oy, s.ox, s.c4, s.cin, s.xcs = k, k, -((k - 1) // 2), -((k - 1) // 2), c4, cin, c4
    return g


def plan(g):
    L.load()
    p = L.QPlan()
    rc = L.query('cat_qconv_plan', C.byref(g), C.byref(p))
    return rc, p


def test_plans_of_the_generator_edge_layers():
    # image stem 3 -> 25 7x7 at batch 16: 16 x 16 tiles (4 pixel groups), all 7 output quads in one wave (8 issued), 4 channels per chunk
    rc, p = plan(geom(16, 256, 256, 256, 256, 3, 25, 7))
    assert rc == 0 and (p.cs, p.nq, p.nsplit, p.th, p.tw) == (4, 8, 1, 16, 16) and p.tiles == 256
    # 49 taps x 1 channel quad = 49 micro steps -> 13 steps (+ 1 spare), 2 filter registers of 256 floats per step and split, table of 15 rows
    assert p.pack_floats == 64 + 14 * 1 * 2 * 256
    # stride 2: 8 x 16 tiles, the output quads split over two waves; the whole K (28 channels) in chunks of 12 (LDS budget of the 17 x 34 patch)
    rc, p = plan(geom(16, 256, 256, 128, 128, 25, 40, 3, stride=2))
    assert rc == 0 and (p.nq, p.nsplit, p.th) == (5, 2, 8) and p.cs == 12 and p.tiles == 16 * 8
    # ConvTranspose2d as four sub-pixel classes: entries = tiles x 4
    rc, p = plan(geom(16, 64, 64, 64, 64, 77, 38, 3, ncls=4))
    assert rc == 0 and p.th == 8 and p.tiles == 8 * 4 * 4
    # few tiles: narrow outputs stay on 8 x 16 tiles (the chip would not fill with 16 x 16)
    rc, p = plan(geom(1, 64, 64, 64, 64, 3, 16, 7))
    assert rc == 0 and p.th == 8 and p.nsplit == 2
    # wide outputs: N blocks over the grid, at most 12 quads per wave
    rc, p = plan(geom(1, 16, 16, 16, 16, 42, 256, 3))
    assert rc == 0 and p.nq <= 12 and p.nq * p.nsplit * 4 >= 256


def test_tile_threshold_hook_round_trips():
    L.load()
    old = L.query('cat_qconv_min_tiles16', -1)
    assert L.query('cat_qconv_min_tiles16', 1) == old
    rc, p = plan(geom(1, 64, 64, 64, 64, 3, 16, 7))
    assert rc == 0 and p.th == 16
    assert L.query('cat_qconv_min_tiles16', old) == 1
    assert L.query('cat_qconv_min_tiles16', -1) == old


@pytest.mark.parametrize('bad', ['stride', 'taps', 'channels', 'classes'])
def test_unsupported_geometries_are_refused(bad):
    g = geom(2, 32, 32, 32, 32, 16, 16, 3)
    if bad == 'stride':
        g.S = 3
    elif bad == 'taps':
        g.seg[0].kh = g.seg[0].kw = 8
    elif bad == 'channels':
        g.seg[0].c4 = 18
    else:
        g.ncls = 4
    rc, _ = plan(g)
    assert rc != 0
    assert b'qconv plan' in L.load().cat_hip_last_error()
