"""torch.autograd.Function wrappers over the C-ABI kernels (include/cat_hip.h).

Tensors on the hot path are "NHWC activations": logical shape [N, C, H, W] (what the reference sees), physical
layout [N][H][W][cs] with cs = round_up(C, 4) floats per pixel and zero padding channels.  PyTorch only provides
device memory, the stream and the autograd tape; every arithmetic op below is a HIP kernel of libcat_hip.so and
raises if the library is missing (no eager / CPU fallback)."""
import ctypes as C
import os
import warnings

import torch

from . import _lib as L
from ._lib import ConvGeom, NormGeom

STATS = {'conform_copies': 0}   # non-native layout fix-ups; must stay 0 on the distillation hot path
_STRICT_LAYOUT = os.environ.get('CAT_STRICT_LAYOUT', '0') == '1'     # debugging aid: raise where a fix-up would happen


# ---------------------------------------------------------------------------------------------- layout helpers
def cs_for(c):
    return (c + 3) // 4 * 4


def empty_act(n, c, h, w, device, cs=None):
    cs = cs or cs_for(c)
    buf = torch.empty((n, h, w, cs), device=device, dtype=torch.float32)
    return buf[..., :c].permute(0, 3, 1, 2)


def act_cs(t):
    n, c, h, w = t.shape
    if w > 1:
        return t.stride(3)
    if h > 1:
        return t.stride(2)
    return t.stride(0) if n > 1 else cs_for(c)


def is_act(t):
    if t.dim() != 4 or t.dtype != torch.float32:
        return False
    n, c, h, w = t.shape
    cs = act_cs(t)
    if cs % 4 or cs < c or t.data_ptr() % 16:
        return False
    ok = (c == 1 or t.stride(1) == 1)
    ok = ok and (w == 1 or t.stride(3) == cs) and (h == 1 or t.stride(2) == w * cs) and (n == 1 or t.stride(0) == h * w * cs)
    return ok


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError('cat_amd ops run on the GPU only (HIP kernels, no CPU fallback); got a %s tensor' % t.device)


_WS = {}


def workspace(nbytes, device):
    """Per-(device, stream) scratch reused by consecutive kernels of one stream."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20) + 3) // 4, device=device, dtype=torch.float32)
        _WS[key] = buf
    return buf


# ---------------------------------------------------------------------------------------------- branch-level concurrency
# The 6 branches of an InvertedResidualChannels block are independent chains of small kernels (64x64 pixels x 6..42 channels): each
# one alone cannot fill 256 CUs.  Running them on separate HIP streams lets the hardware overlap their launches, tails and
# latency-bound phases; autograd replays every backward node on the stream its forward ran on, so the backward overlaps too.
# Round 6, measured (profiles/r06_tconv_ab.txt section 7): since the blocks became a handful of chip-filling launches, the fork / join edges cost
# the pix2pix / CycleGAN-style step 1.5 - 3 % (321 images/s on ONE stream against 313 - 316), while the GauGAN step -- ~1 500 small launches --
# still gains 6 - 9 % from them.  The distillers therefore pick the default (default_branch_streams: inception off, SPADE on); CAT_BRANCH_STREAMS
# set in the environment overrides both.
_SIDE = {}
_BRANCH_ENV = os.environ.get('CAT_BRANCH_STREAMS')
_BRANCH_STREAMS = _BRANCH_ENV != '0'


def branch_streams_enabled():
    return _BRANCH_STREAMS


def set_branch_streams(on):
    global _BRANCH_STREAMS
    _BRANCH_STREAMS = bool(on)


def default_branch_streams(on):
    """A model's preference (called from the distillers' constructors); an explicit CAT_BRANCH_STREAMS wins."""
    if _BRANCH_ENV is None:
        set_branch_streams(on)


class _BranchOutFn(torch.autograd.Function):
    """Identity at the end of a side-stream branch.  Its backward runs on that side stream and receives a gradient that was
    allocated on the MAIN stream (the branch sum's gradient, shared by all branches): mark it as in use here, otherwise the caching
    allocator may hand the block to the next main-stream kernel while this branch is still reading it."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        dy.record_stream(torch.cuda.current_stream())
        return dy


_USED = {}


def sync_side_streams():
    """Make the current stream wait for every side stream that ran branch work since the last call.  Must follow each backward
    pass: weight-gradient kernels write straight into the optimiser's flat buffer from the branch streams, and a branch whose input
    needs no gradient (the SPADE gamma/beta branches read the segmentation map) ends there -- nothing downstream would otherwise
    order the optimiser step (or a captured graph's next node) after them."""
    main = torch.cuda.current_stream()
    used = _USED.pop((main.device, main.cuda_stream), None)
    if used:
        for st in used:
            main.wait_stream(st)


def run_on_side_streams(fns, inputs):
    main = torch.cuda.current_stream()
    dev = inputs[0].device
    key = (dev, main.cuda_stream)
    _USED.setdefault(key, set()).update(_SIDE.get(key, [])[:len(fns)])
    pool = _SIDE.get(key)
    if pool is None or len(pool) < len(fns):
        pool = [torch.cuda.Stream(device=dev) for _ in range(max(len(fns), 6))]
        _SIDE[key] = pool
    _USED[key].update(pool[:len(fns)])
    fork = main.record_event()
    outs = []
    for fn, xi, st in zip(fns, inputs, pool):
        st.wait_event(fork)
        xi.record_stream(st)
        with torch.cuda.stream(st):
            o = fn(xi)
            if o.requires_grad:
                o = _BranchOutFn.apply(o)
        o.record_stream(main)
        main.wait_event(st.record_event())
        outs.append(o)
    return outs


class SideJobs:
    """Independent kernel sequences of one backward node (weight-gradient launches of a fused block) dealt round-robin to the side
    streams: `fork()` after their inputs are enqueued on the current stream, `run(fn)` per job, `join()` before the node returns (the
    temporaries the jobs read are freed in current-stream order after that)."""

    def __init__(self, device, width=6):
        self.main = torch.cuda.current_stream()
        self.on = _BRANCH_STREAMS
        if self.on:
            key = (device, self.main.cuda_stream)
            pool = _SIDE.get(key)
            if pool is None or len(pool) < width:
                pool = [torch.cuda.Stream(device=device) for _ in range(max(width, 6))]
                _SIDE[key] = pool
            self.pool, self.used, self.k, self.width = pool, set(), 0, width

    def fork(self):
        self.ev = self.main.record_event() if self.on else None

    def run(self, fn):
        if not self.on:
            return fn()
        s = self.pool[self.k % self.width]
        self.k += 1
        if s not in self.used:
            s.wait_event(self.ev)
            self.used.add(s)
        with torch.cuda.stream(s):
            fn()

    def refork(self):
        """Later jobs depend on work enqueued on the main stream since fork()."""
        if self.on:
            self.ev = self.main.record_event()
            for s in self.pool[:self.width]:
                s.wait_event(self.ev)
                self.used.add(s)

    def join(self):
        if self.on:
            for s in self.used:
                self.main.wait_event(s.record_event())
            self.used = set()


def to_nhwc(x):
    """NCHW-contiguous (reference layout) -> NHWC activation, through cat_nchw_to_nhwc."""
    _require_cuda(x)
    if is_act(x):
        return x
    x = x.contiguous()
    n, c, h, w = x.shape
    y = empty_act(n, c, h, w, x.device)
    L.call('cat_nchw_to_nhwc', _p(x), _p(y), n, c, h, w, act_cs(y), _stream())
    return y


def to_nchw(x):
    """NHWC activation -> NCHW-contiguous tensor."""
    _require_cuda(x)
    if not is_act(x):
        return x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    L.call('cat_nhwc_to_nchw', _p(x), _p(y), n, c, h, w, act_cs(x), _stream())
    return y


def conform(t):
    """Make `t` an NHWC activation.  Gradients produced by our own kernels already are; anything else (e.g. a
    gradient torch itself accumulated) is re-laid-out and counted."""
    if is_act(t):
        return t
    STATS['conform_copies'] += 1
    if _STRICT_LAYOUT:
        raise RuntimeError(f'non-native activation layout: shape {tuple(t.shape)}, strides {t.stride()}, dtype {t.dtype}')
    if t.is_contiguous():
        return to_nhwc(t)
    n, c, h, w = t.shape
    y = empty_act(n, c, h, w, t.device)
    full = torch.as_strided(y, (n, act_cs(y), h, w), y.stride())
    full.zero_()
    y.copy_(t)
    return y


def padded_weight_like(shape, device):
    """Storage for a conv weight of logical shape [O, I, kh, kw]: physical [O][kh][kw][round_up(I, 4)], zero padded."""
    o, i, kh, kw = shape
    buf = torch.zeros((o, kh, kw, cs_for(i)), device=device, dtype=torch.float32)
    return buf[..., :i].permute(0, 3, 1, 2)


def weight_wcs(w):
    """Floats per (cout, tap) if `w` ([O, I, kh, kw]) is laid out [O][kh][kw][wcs] (dense channels_last or padded), else None."""
    if w.dim() != 4 or w.dtype != torch.float32 or w.data_ptr() % 16:
        return None
    o, i, kh, kw = w.shape
    s0, s1, s2, s3 = w.stride()
    wcs = s3 if kw > 1 else (s2 if kh > 1 else (s0 if o > 1 else cs_for(i)))
    if o == 1 and kh == 1 and kw == 1 and i == 1:
        return 1      # a single element: every row stride describes it
    if o == 1 and kh == 1 and kw == 1 and wcs != i:
        # a single [1][I] row with I % 4 != 0: only the strides of the size-1 dims tell a padded row (padded_weight_like / FusedAdam views keep
        # s0 = s2 = s3 = round_up(I, 4)) from a dense one (a test tensor, a checkpoint tensor, a slice of a larger storage: s3 = 1 or I) -- never
        # inferred from what happens to lie behind the row in its storage.  Dense rows are re-laid-out by weight_cl
        if not (s0 == wcs and s2 == wcs and s3 == wcs) or w.untyped_storage().nbytes() // 4 - w.storage_offset() < wcs:
            return None
    ok = wcs >= i and (i == 1 or s1 == 1) and (kw == 1 or s3 == wcs) and (kh == 1 or s2 == kw * wcs) and (o == 1 or s0 == kh * kw * wcs)
    return wcs if ok else None


def weight_cl(w):
    """(tensor, wcs): the weight in kernel layout.  Parameters owned by cat_amd modules / FusedAdam already are (padded
    channels_last); anything else (a test tensor, a freshly loaded NCHW checkpoint tensor) is re-laid-out once."""
    wcs = weight_wcs(w)
    if wcs is not None:
        return w, wcs
    out = padded_weight_like(w.shape, w.device)
    out.copy_(w.detach())
    return out, weight_wcs(out)


def _conv_geom(n, h, w, cin, xcs, ho, wo, cout, ycs, kh, kw, stride, pad, pad_mode, act=0, slope=0.0, ycw=0, wcs=0):
    return ConvGeom(n, h, w, cin, xcs, ho, wo, cout, ycs, kh, kw, stride, pad, pad_mode, act, slope, ycw, wcs)


def _grad_target(param):
    """FusedAdam registers a flat gradient buffer view on its parameters: wgrad kernels then write (or
    accumulate) straight into it and autograd sees no weight gradient at all."""
    return getattr(param, '_cat_grad_view', None)


def _write_param_grad(param, kernel):
    """kernel(dst_tensor, accumulate_flag).  Returns the tensor to hand back to autograd (None if direct)."""
    tgt = _grad_target(param)
    if tgt is not None:
        st = param._cat_grad_state
        kernel(tgt, 0 if st['fresh'] else 1)
        st['fresh'] = False
        return None
    if param.dim() == 4 and param.shape[1] > 1:
        g = padded_weight_like(param.shape, param.device)
    else:
        g = torch.empty(param.shape, device=param.device, dtype=param.dtype)
    kernel(g, 0)
    return g


class WgradBatch:
    """Weight gradients of one backward node collected and issued together (cat_conv2d_wgrad_batch): the producer kernels back to back, their
    partial sums reduced by ONE launch instead of one 5 - 8 us launch each.  add(param, make) with make(dst) -> (ConvGeom, x pointer, dy pointer)
    picks the destination like _write_param_grad (the optimizer's flat gradient view, or a fresh tensor returned through `grads`); flush()
    launches.  The tensors the pointers address must stay alive until flush() -- the caller's frame holds them."""

    def __init__(self, device, grads):
        self.device, self.grads, self.items = device, grads, []

    def add(self, param, make):
        self.grads[id(param)] = _write_param_grad(param, lambda dst_, acc: self.items.append((make(dst_), dst_, acc)))

    def add_into(self, dst, acc, geom, xptr, dyptr):
        self.items.append(((geom, xptr, dyptr), dst, acc))

    def flush(self):
        st = _stream()
        for i0 in range(0, len(self.items), L.WGRAD_BATCH_MAX):
            chunk = self.items[i0:i0 + L.WGRAD_BATCH_MAX]
            arr = (L.WgradItem * len(chunk))()
            for it, ((gw, xp, dyp), dst, acc) in zip(arr, chunk):
                C.memmove(C.byref(it.g), C.byref(gw), C.sizeof(L.ConvGeom))
                it.x = xp.value if isinstance(xp, C.c_void_p) else xp
                it.dy = dyp.value if isinstance(dyp, C.c_void_p) else dyp
                it.dw, it.accumulate = dst.data_ptr(), acc
            ws = workspace(int(L.query('cat_conv2d_wgrad_batch_ws_bytes', arr, len(chunk))), self.device)
            L.call('cat_conv2d_wgrad_batch', arr, len(chunk), _p(ws), st)
        self.items = []


def _act_bwd(y, dy, act, slope):
    n, c, h, w = y.shape
    dz = empty_act(n, c, h, w, y.device, act_cs(y))
    L.call('cat_act_bwd', _p(y), _p(dy), _p(dz), n * h * w * act_cs(y), act, slope, _stream())
    return dz


def _nchw(t):
    n, c, h, w = t.shape
    return n, c, h, w


# ---------------------------------------------------------------------------------------------- convolutions
_TCONV = os.environ.get('CAT_TCONV', '1') != '0'      # LDS-tile kernels with packed filters for stride-1 3x3 / 5x5 layers (A/B switch)
_TCONV_MIN_TILES = int(os.environ.get('CAT_TCONV_MIN_TILES', '96'))


def set_tconv_min_tiles(n):
    """Fewest 8 x 16 output tiles for which the LDS-tile kernels (and the fused block path built on them) are chosen.  The default keeps
    planes that cannot fill the chip on the split-K im2col kernels; tests lower it to drive small batches through the same kernels."""
    global _TCONV_MIN_TILES
    old, _TCONV_MIN_TILES = _TCONV_MIN_TILES, int(n)
    return old


def tconv_applicable(n, h, w, cout, kh, kw, stride, pad, cin=None):
    """Stride-1 "same" 3x3 / 5x5 convolutions with enough output tiles to fill the chip go through csrc/conv_pk.hip (forward and
    input gradient); tiny planes (the SPADE generators' 4x8 .. 16x32 stages) keep the split-K im2col kernels, Cout <= 3 the direct ones.
    Wide layers the direct-to-LDS 128 x 128 x 32 GEMM tiles accept (reduction channels % 32 == 0, more than 96 output channels: e.g. the
    256 -> 256 3x3 convs of a resnet generator) stay on those: their matrix pipe runs 0.9 busy against 0.75-0.8 for the LDS-tile kernel."""
    if not _TCONV or stride != 1 or kh != kw or kh not in (3, 5) or pad != (kh - 1) // 2 or cout <= 3:
        return False
    if cin is not None and cin % 32 == 0 and cout > 96:
        return False
    return n * ((h + 7) // 8) * ((w + 15) // 16) >= _TCONV_MIN_TILES


def packed_filter(weight, wcl, mode):
    """The conv weight in the MFMA-group order cat_tconv_fwd consumes, cached on the tensor and re-packed (in place) when the
    weight changed: torch's version counter for ordinary tensors, the optimizer epoch for FusedAdam-owned parameters (updated
    through raw pointers).  One ~2 us launch per layer and direction per step."""
    from . import optim, tconv
    owner = weight if weight is not None else wcl
    if owner is not wcl and owner.data_ptr() != wcl.data_ptr():      # a re-laid-out temporary (foreign layout): nothing to key a cache on
        return tconv.pack(wcl, mode)
    trainable = getattr(owner, '_cat_grad_view', None) is not None
    key = (wcl.data_ptr(), wcl._version, optim.weights_epoch() if trainable else -1)
    cache = getattr(owner, '_cat_pk', None)
    if cache is None:
        cache = owner._cat_pk = {}
    ent = cache.get(mode)
    if ent is not None and ent[0] == key:
        return ent[1]
    if ent is None or ent[1].device != wcl.device:
        o, i, kh, kw = wcl.shape
        nn, ck = (o, i) if mode == tconv.FWD else (i, o)
        ent = [None, torch.empty(tconv.pack_floats(kh, tconv.cs4(ck), nn), device=wcl.device, dtype=torch.float32)]
        cache[mode] = ent
    tconv.pack_into(ent[1], wcl, mode)
    ent[0] = key
    return ent[1]


def _conv_fwd(g, x, w, bias, y, st):
    """cat_conv2d_fwd, or its split-K variant when the layer's tile grid is too small for the chip."""
    nb = L.query('cat_conv2d_fwd_ws_bytes', C.byref(g))
    if nb:
        L.call('cat_conv2d_fwd_ws', C.byref(g), _p(x), _p(w), _p(bias), _p(y), _p(workspace(nb, y.device)), st)
    else:
        L.call('cat_conv2d_fwd', C.byref(g), _p(x), _p(w), _p(bias), _p(y), st)


def transposed_filter(weight, wcl, g):
    """[Cin][kh][kw][Cout] copy of a conv weight for cat_conv2d_dgrad_t, cached on the tensor and refreshed when the weight changed (same
    keying as packed_filter: version counter, or the optimizer epoch for FusedAdam-owned parameters)."""
    from . import optim
    owner = weight if weight is not None else wcl
    trainable = getattr(owner, '_cat_grad_view', None) is not None
    key = (wcl.data_ptr(), wcl._version, optim.weights_epoch() if trainable else -1)
    ent = getattr(owner, '_cat_wt', None)
    if ent is not None and ent[0] == key:
        return ent[1]
    o, i, kh, kw = wcl.shape
    buf = ent[1] if (ent is not None and ent[1].numel() == o * i * kh * kw and ent[1].device == wcl.device) else \
        torch.empty(o * i * kh * kw, device=wcl.device, dtype=torch.float32)
    L.call('cat_conv2d_weight_transpose', C.byref(g), _p(wcl), _p(buf), _stream())
    owner._cat_wt = (key, buf)
    return buf


# ---------------------------------------------------------------------------------------------- opt-in split-bf16 backward tiles
# CAT_MFMA=bf16x3: the wide 4x4 PatchGAN layers' data / weight gradients on the bf16 matrix pipe in three-product split form (csrc/conv_split.hip;
# numerically a no-op for the BACKWARD tiles at the parity suite's bars, not for the forward).  Never the default.
_MFMA_SPLIT = os.environ.get('CAT_MFMA', '') == 'bf16x3'


def set_mfma_split(on):
    global _MFMA_SPLIT
    old, _MFMA_SPLIT = _MFMA_SPLIT, bool(on)
    return old


def split_planes(t, n=None):
    """fp32 tensor (dense storage of n floats, n % 4 == 0) -> its two bf16 planes [2][n] (cat_split_bf16)"""
    n = t.numel() if n is None else n
    out = torch.empty(2 * n, device=t.device, dtype=torch.bfloat16)
    L.call('cat_split_bf16', _p(t), _p(out), n, _stream())
    return out


def split_transposed_filter(weight, wcl, g):
    """bf16 planes of the [Cin][kh][kw][Cout] filter copy, cached like transposed_filter"""
    wt = transposed_filter(weight, wcl, g)
    owner = weight if weight is not None else wcl
    key = owner._cat_wt[0]
    ent = getattr(owner, '_cat_wtp', None)
    if ent is not None and ent[0] == key:
        return ent[1]
    planes = split_planes(wt)
    owner._cat_wtp = (key, planes)
    return planes


def _conv_dgrad(g, dy, w, bias, dx, dxcs, dxcw, st, weight=None):
    if L.query('cat_conv2d_dgrad_t_applicable', C.byref(g)) and weight_wcs(w) is not None:
        wt = transposed_filter(weight, w, g)
        L.call('cat_conv2d_dgrad_t', C.byref(g), _p(dy), _p(w), _p(wt), _p(bias), _p(dx), dxcs, dxcw, st)
        return
    nb = L.query('cat_conv2d_dgrad_ws_bytes', C.byref(g), dxcs)
    if nb:
        L.call('cat_conv2d_dgrad_ws', C.byref(g), _p(dy), _p(w), _p(bias), _p(dx), dxcs, dxcw, _p(workspace(nb, dx.device)), st)
    else:
        L.call('cat_conv2d_dgrad', C.byref(g), _p(dy), _p(w), _p(bias), _p(dx), dxcs, dxcw, st)


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (+ preceding ReflectionPad2d, + following pointwise activation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, pad_mode, act, slope, q=None):
        """q (optional): {'layer': qconv.Layer, 'stats': bool} -- run the layer on the quad-granule kernel (csrc/conv_q.hip); with 'stats' the
        per-tile statistics of the output land in q['table'] (+ q['plan']) for the train-mode norm behind (NormActFn `tiles`)."""
        _require_cuda(x)
        x = conform(x)
        wcl, wcs = weight_cl(weight)
        n, cin, h, w = x.shape
        cout, cin_w, kh, kw = weight.shape
        if cin_w != cin:
            raise RuntimeError(f'conv2d: input has {cin} channels, weight expects {cin_w}')
        ho = (h + 2 * pad - kh) // stride + 1
        wo = (w + 2 * pad - kw) // stride + 1
        y = empty_act(n, cout, ho, wo, x.device)
        if q is not None:
            run_qconv(q, x, bias, y, act, slope)
        elif tconv_applicable(n, h, w, cout, kh, kw, stride, pad, cin):
            from . import tconv
            pk = packed_filter(weight, wcl, tconv.FWD)
            tconv.run([tconv.Segment(x, kh, pad, pad_mode == L.PAD_REFLECT, 0)], pk, bias, y, cout, n, h, w, ho, wo, act, slope)
        else:
            g = _conv_geom(n, h, w, cin, act_cs(x), ho, wo, cout, act_cs(y), kh, kw, stride, pad, pad_mode, act, slope, act_cs(y), wcs)
            _conv_fwd(g, x, wcl, bias, y, _stream())
        ctx.geom = (n, h, w, cin, act_cs(x), ho, wo, cout, act_cs(y), kh, kw, stride, pad, pad_mode, wcs)
        ctx.act, ctx.slope = act, slope
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, wcl, y if act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wcl, y = ctx.saved_tensors
        n, h, w, cin, xcs, ho, wo, cout, ycs, kh, kw, stride, pad, pad_mode, wcs = ctx.geom
        dy = conform(dy)
        if ctx.act != L.ACT_NONE:
            dy = _act_bwd(y, dy, ctx.act, ctx.slope)
        g = _conv_geom(n, h, w, cin, xcs, ho, wo, cout, act_cs(dy), kh, kw, stride, pad, pad_mode, wcs=wcs)
        st = _stream()
        dx = dw = db = None
        dyp = None      # split-bf16 planes of dy, shared by the data and the weight gradient
        if ctx.needs_input_grad[0]:
            tile = tconv_applicable(n, h, w, cin, kh, kw, stride, pad, cout)
            if tile:
                from . import tconv
                pk = packed_filter(ctx.weight, wcl, tconv.DGRAD)
            if pad_mode == L.PAD_REFLECT and pad > 0:
                dxp = empty_act(n, cin, h + 2 * pad, w + 2 * pad, x.device)
                if tile:   # gradient of the PADDED plane: full correlation of dy with the flipped filters
                    tconv.run([tconv.Segment(dy, kh, kh - 1, False, 0)], pk, None, dxp, cin, n, ho, wo, h + 2 * pad, w + 2 * pad)
                else:
                    _conv_dgrad(g, dy, wcl, None, dxp, act_cs(dxp), act_cs(dxp), st, ctx.weight)
                dx = empty_act(n, cin, h, w, x.device)
                L.call('cat_reflect_pad_bwd', _p(dxp), _p(dx), n, h, w, cin, act_cs(dx), pad, st)
            else:
                dx = empty_act(n, cin, h, w, x.device)
                if tile:
                    tconv.run([tconv.Segment(dy, kh, kh - 1 - pad, False, 0)], pk, None, dx, cin, n, ho, wo, h, w)
                elif _MFMA_SPLIT and act_cs(dx) == cin and weight_wcs(wcl) is not None and L.query('cat_conv2d_dgrad_split_applicable', C.byref(g)):
                    dyp = split_planes(dy)
                    L.call('cat_conv2d_dgrad_split', C.byref(g), _p(dyp), _p(split_transposed_filter(ctx.weight, wcl, g)), _p(dx), act_cs(dx), st)
                    STATS['split_dgrad'] = STATS.get('split_dgrad', 0) + 1
                else:
                    _conv_dgrad(g, dy, wcl, None, dx, act_cs(dx), act_cs(dx), st, ctx.weight)
        if ctx.needs_input_grad[1]:
            ws = workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)), x.device)

            def k(dst, acc):
                nonlocal dyp
                gw = _conv_geom(n, h, w, cin, xcs, ho, wo, cout, act_cs(dy), kh, kw, stride, pad, pad_mode, wcs=_grad_wcs(dst))
                if _MFMA_SPLIT and L.query('cat_conv2d_wgrad_split_applicable', C.byref(gw)):
                    if dyp is None:
                        dyp = split_planes(dy)
                    ws2 = workspace(L.query('cat_conv2d_wgrad_split_ws_bytes', C.byref(gw)), x.device)
                    L.call('cat_conv2d_wgrad_split', C.byref(gw), _p(split_planes(x)), _p(dyp), _p(dst), acc, _p(ws2), st)
                    STATS['split_wgrad'] = STATS.get('split_wgrad', 0) + 1
                    return
                L.call('cat_conv2d_wgrad', C.byref(gw), _p(x), _p(dy), _p(dst), acc, _p(ws), st)
            dw = _write_param_grad(ctx.weight, k)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            m = n * ho * wo
            ws = workspace(L.query('cat_channel_sum_ws_bytes', m, act_cs(dy)), x.device)
            db = _write_param_grad(ctx.bias, lambda dst, acc: L.call('cat_channel_sum', _p(dy), m, cout, act_cs(dy), _p(dst), acc,
                                                                     _p(ws), st))
        return dx, dw, db, None, None, None, None, None, None


def run_qconv(q, x, bias, y, act=L.ACT_NONE, slope=0.0, pre=None):
    """One launch of a qconv.Layer; q['stats'] -> allocate the tile-statistics table and leave (table, plan) in q."""
    layer = q['layer']
    table, scs = None, 0
    if q.get('stats'):
        n = x.shape[0]
        scs = act_cs(y)
        plan0 = layer.plan_for(x)
        table = torch.empty(n * plan0.tiles * 2 * scs, device=x.device, dtype=torch.float32)
    plan = layer.run(x, bias, y, act=act, slope=slope, pre=pre, stats=table, scs=scs)
    q['table'], q['plan'], q['scs'] = table, plan, scs
    return plan


class Normed:
    """A pre-norm conv output with its pending train-mode norm + activation (scale / shift rows from cat_tnorm_finalize2): produced only in
    no-grad forwards, consumed by a quad-granule conv that applies them while staging (no normalised copy is written), or materialised."""
    __slots__ = ('z', 'scale', 'shift', 'groups', 'act', 'slope')

    def __init__(self, z, scale, shift, groups, act, slope):
        self.z, self.scale, self.shift, self.groups, self.act, self.slope = z, scale, shift, groups, act, slope

    @property
    def shape(self):
        return self.z.shape

    def pre(self):
        """(scale, shift, sstride, act, slope) for qconv.Layer.run"""
        return self.scale, self.shift, (self.scale.shape[1] if self.groups > 1 else 0), self.act, self.slope

    def materialize(self):
        z = self.z
        n, c, h, w = z.shape
        cs = act_cs(z)
        y = empty_act(n, c, h, w, z.device, cs)
        g = self.groups
        L.call('cat_affine_res_fwd', _p(z), cs, _p(self.scale), _p(self.shift), self.scale.shape[1] if g > 1 else 0, None, 0, _p(y), cs, g,
               (n // g) * h * w, cs, self.act, self.slope, _stream())
        return y


def _grad_wcs(t):
    wcs = weight_wcs(t)
    if wcs is None:
        raise RuntimeError('conv weight gradient buffer must be laid out [O][kh][kw][wcs] (channels_last, optionally padded)')
    return wcs


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d: forward is the dgrad kernel of the equivalent strided conv
    (conv input = our output), backward-data is that conv's forward, backward-weights its wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, output_padding, act=0, slope=0.0):
        _require_cuda(x)
        x = conform(x)
        wcl, wcs = weight_cl(weight)
        n, cin_t, hi, wi = x.shape
        cin_w, cout_t, kh, kw = weight.shape
        if cin_w != cin_t:
            raise RuntimeError(f'conv_transpose2d: input has {cin_t} channels, weight expects {cin_w}')
        if act != L.ACT_NONE and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
            raise NotImplementedError('conv_transpose2d: the fused epilogue activation is the frozen (no-grad) path only')
        ho = (hi - 1) * stride - 2 * pad + kh + output_padding
        wo = (wi - 1) * stride - 2 * pad + kw + output_padding
        y = empty_act(n, cout_t, ho, wo, x.device)
        # equivalent conv: input (ho, wo, cout_t) -> output (hi, wi, cin_t)
        g = _conv_geom(n, ho, wo, cout_t, act_cs(y), hi, wi, cin_t, act_cs(x), kh, kw, stride, pad, L.PAD_ZERO, act, slope, wcs=wcs)
        L.call('cat_conv2d_dgrad', C.byref(g), _p(x), _p(wcl), _p(bias), _p(y), act_cs(y), act_cs(y), _stream())
        ctx.geom = (n, ho, wo, cout_t, act_cs(y), hi, wi, cin_t, act_cs(x), kh, kw, stride, pad, wcs)
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, wcl)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wcl = ctx.saved_tensors
        n, ho, wo, cout_t, ycs, hi, wi, cin_t, xcs, kh, kw, stride, pad, wcs = ctx.geom
        dy = conform(dy)
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = empty_act(n, cin_t, hi, wi, x.device)
            g = _conv_geom(n, ho, wo, cout_t, act_cs(dy), hi, wi, cin_t, act_cs(dx), kh, kw, stride, pad, L.PAD_ZERO, 0, 0.0, act_cs(dx), wcs)
            L.call('cat_conv2d_fwd', C.byref(g), _p(dy), _p(wcl), None, _p(dx), st)
        if ctx.needs_input_grad[1]:
            g = _conv_geom(n, ho, wo, cout_t, act_cs(dy), hi, wi, cin_t, xcs, kh, kw, stride, pad, L.PAD_ZERO)
            ws = workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(g)), x.device)

            def k(dst, acc):
                gw = _conv_geom(n, ho, wo, cout_t, act_cs(dy), hi, wi, cin_t, xcs, kh, kw, stride, pad, L.PAD_ZERO, wcs=_grad_wcs(dst))
                L.call('cat_conv2d_wgrad', C.byref(gw), _p(dy), _p(x), _p(dst), acc, _p(ws), st)
            dw = _write_param_grad(ctx.weight, k)
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            m = n * ho * wo
            ws = workspace(L.query('cat_channel_sum_ws_bytes', m, act_cs(dy)), x.device)
            db = _write_param_grad(ctx.bias, lambda dst, acc: L.call('cat_channel_sum', _p(dy), m, cout_t, act_cs(dy), _p(dst), acc,
                                                                     _p(ws), st))
        return dx, dw, db, None, None, None, None, None


class DwConv2dFn(torch.autograd.Function):
    """Depthwise nn.Conv2d(groups=C), stride 1 (+ preceding ReflectionPad2d)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, pad_mode):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        cw, one, kh, kw = weight.shape
        if cw != c or one != 1:
            raise RuntimeError('depthwise conv2d: weight must be [C,1,kh,kw] with C == input channels')
        wc = weight.contiguous()
        ho, wo = h + 2 * pad - kh + 1, w + 2 * pad - kw + 1
        y = empty_act(n, c, ho, wo, x.device)
        g = _conv_geom(n, h, w, c, act_cs(x), ho, wo, c, act_cs(y), kh, kw, 1, pad, pad_mode)
        L.call('cat_dwconv2d_fwd', C.byref(g), _p(x), _p(wc), _p(bias), _p(y), _stream())
        ctx.geom = (n, h, w, c, act_cs(x), ho, wo, kh, kw, pad, pad_mode)
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, wc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        n, h, w, c, xcs, ho, wo, kh, kw, pad, pad_mode = ctx.geom
        dy = conform(dy)
        g = _conv_geom(n, h, w, c, xcs, ho, wo, c, act_cs(dy), kh, kw, 1, pad, pad_mode)
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if pad_mode == L.PAD_REFLECT and pad > 0:
                dxp = empty_act(n, c, h + 2 * pad, w + 2 * pad, x.device)
                L.call('cat_dwconv2d_dgrad', C.byref(g), _p(dy), _p(wc), _p(dxp), act_cs(dxp), st)
                dx = empty_act(n, c, h, w, x.device)
                L.call('cat_reflect_pad_bwd', _p(dxp), _p(dx), n, h, w, c, act_cs(dx), pad, st)
            else:
                dx = empty_act(n, c, h, w, x.device)
                L.call('cat_dwconv2d_dgrad', C.byref(g), _p(dy), _p(wc), _p(dx), act_cs(dx), st)
        if ctx.needs_input_grad[1]:
            ws = workspace(L.query('cat_dwconv2d_wgrad_ws_bytes', C.byref(g)), x.device)
            dw = _write_param_grad(ctx.weight, lambda dst, acc: L.call('cat_dwconv2d_wgrad', C.byref(g), _p(x), _p(dy), _p(dst), acc,
                                                                       _p(ws), st))
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            m = n * ho * wo
            ws = workspace(L.query('cat_channel_sum_ws_bytes', m, act_cs(dy)), x.device)
            db = _write_param_grad(ctx.bias, lambda dst, acc: L.call('cat_channel_sum', _p(dy), m, c, act_cs(dy), _p(dst), acc, _p(ws),
                                                                     st))
        return dx, dw, db, None, None


# ---------------------------------------------------------------------------------------------- normalisation
class NormActFn(torch.autograd.Function):
    """InstanceNorm2d / BatchNorm2d with batch statistics, fused with the activation behind it."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, mode, eps, momentum, act, slope, num_batches=None, tiles=None):
        """tiles (optional): the producing conv's per-tile statistics {'table', 'plan', 'scs', 'lat': (h, w), 'ncls'} (cat_qconv_fwd `stats`):
        the statistics pass is skipped -- cat_tnorm_finalize2 + one apply pass."""
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        cs = act_cs(x)
        y = empty_act(n, c, h, w, x.device, cs)
        g = NormGeom(n, h * w, c, cs, mode, eps, momentum, act, slope)
        groups = n if mode == L.NORM_INSTANCE else 1
        mean = torch.empty((groups, c), device=x.device, dtype=torch.float32)
        rstd = torch.empty((groups, c), device=x.device, dtype=torch.float32)
        if tiles is not None:
            scale, shift = norm_from_tiles(tiles, n, c, groups, gamma, beta, running_mean, running_var, num_batches, eps, momentum, mean, rstd)
            L.call('cat_affine_res_fwd', _p(x), cs, _p(scale), _p(shift), scale.shape[1] if groups > 1 else 0, None, 0, _p(y), cs, groups,
                   (n // groups) * h * w, cs, act, slope, _stream())
        else:
            ws = workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
            L.call('cat_norm_fwd', C.byref(g), _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                   _p(num_batches), _p(ws), _stream())
        ctx.geom = (n, h * w, c, cs, mode, eps, momentum, act, slope)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = conform(dy)
        if act_cs(dy) != act_cs(x):
            raise RuntimeError('norm backward: gradient pixel stride differs from the input')
        g = NormGeom(*ctx.geom)
        dx = empty_act(n, c, h, w, x.device, act_cs(x))
        ws = workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
        st = _stream()
        need_g = gamma is not None and ctx.needs_input_grad[1]
        need_b = beta is not None and ctx.needs_input_grad[2]
        dgamma = dbeta = None
        tg = _grad_target(ctx.gamma) if need_g else None
        tb = _grad_target(ctx.beta) if need_b else None
        direct = need_g and need_b and tg is not None and tb is not None
        if direct:
            sg, sb = ctx.gamma._cat_grad_state, ctx.beta._cat_grad_state
            if sg['fresh'] != sb['fresh']:
                raise RuntimeError('norm backward: gamma / beta gradient buffers out of sync')
            acc = 0 if sg['fresh'] else 1
            L.call('cat_norm_bwd', C.byref(g), _p(x), _p(dy), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(tg), _p(tb), acc,
                   _p(ws), st)
            sg['fresh'] = sb['fresh'] = False
        else:
            dgamma = torch.empty_like(gamma) if need_g else None
            dbeta = torch.empty_like(beta) if need_b else None
            L.call('cat_norm_bwd', C.byref(g), _p(x), _p(dy), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta), 0,
                   _p(ws), st)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def norm_from_tiles(tiles, n, c, groups, gamma, beta, running_mean, running_var, num_batches, eps, momentum, mean=None, rstd=None):
    """cat_tnorm_finalize2 over a conv's tile-statistics table: scale / shift rows [groups][scs] of the train-mode norm (+ its running
    statistics, torch semantics) and, if given, mean / rstd [groups][c] for cat_norm_bwd."""
    table, plan, scs = tiles['table'], tiles['plan'], tiles['scs']
    lat_h, lat_w = tiles['lat']
    dev = table.device
    scale = torch.empty((groups, scs), device=dev, dtype=torch.float32)
    shift = torch.empty((groups, scs), device=dev, dtype=torch.float32)
    if mean is None:
        mean = torch.empty((groups, c), device=dev, dtype=torch.float32)
        rstd = torch.empty((groups, c), device=dev, dtype=torch.float32)
    sl = (L.NSlice * 1)()
    sl[0].c0, sl[0].c = 0, c
    if running_mean is not None:
        sl[0].running_mean, sl[0].running_var = running_mean.data_ptr(), running_var.data_ptr()
        if num_batches is not None:
            sl[0].num_batches = num_batches.data_ptr()
    L.call('cat_tnorm_finalize2', _p(table), scs, groups, n, lat_h, lat_w, plan.th, plan.tw, tiles.get('ncls', 1), _p(gamma), _p(beta), 1, sl,
           eps, momentum, _p(scale), _p(shift), _p(mean), _p(rstd), c, _stream())
    return scale, shift


def affine_act(x, scale, shift, act, slope):
    """Inference-mode norm (running statistics folded into scale/shift): no autograd (frozen teacher only)."""
    _require_cuda(x)
    x = conform(x)
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError('eval-mode BatchNorm is only implemented for the frozen (no-grad) teacher')
    n, c, h, w = x.shape
    y = empty_act(n, c, h, w, x.device, act_cs(x))
    L.call('cat_affine_act_fwd', _p(x), _p(scale), _p(shift), _p(y), n * h * w, c, act_cs(x), act, slope, _stream())
    return y


def bn_fold(gamma, beta, running_mean, running_var, eps):
    c = running_mean.numel()
    scale = torch.empty(c, device=running_mean.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    L.call('cat_bn_fold', _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, c, _p(scale), _p(shift), _stream())
    return scale, shift


# ---------------------------------------------------------------------------------------------- pointwise
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        y = empty_act(n, c, h, w, x.device, act_cs(x))
        L.call('cat_act_fwd', _p(x), _p(y), n * h * w * act_cs(x), act, slope, _stream())
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _act_bwd(y, conform(dy), ctx.act, ctx.slope), None, None


class ReplicatePadFn(torch.autograd.Function):
    """nn.ReplicationPad2d(pad): materialised (reference inception_modules.py:114-115; no launch script selects it)."""

    @staticmethod
    def forward(ctx, x, pad):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        y = empty_act(n, c, h + 2 * pad, w + 2 * pad, x.device, act_cs(x))
        L.call('cat_replicate_pad_fwd', _p(x), _p(y), n, h, w, c, act_cs(x), pad, _stream())
        ctx.dims = (n, c, h, w, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, pad = ctx.dims
        dy = conform(dy)
        dx = empty_act(n, c, h, w, dy.device, act_cs(dy))
        L.call('cat_replicate_pad_bwd', _p(dy), _p(dx), n, h, w, c, act_cs(dy), pad, _stream())
        return dx, None


class AddNFn(torch.autograd.Function):
    """sum of k same-shaped activations (branch sum + residual of InvertedResidualChannels)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [conform(x) for x in xs]
        _require_cuda(xs[0])
        n, c, h, w = xs[0].shape
        cs = act_cs(xs[0])
        for x in xs:
            if x.shape != xs[0].shape or act_cs(x) != cs:
                raise RuntimeError('add_n: operands must share shape and pixel stride')
        y = empty_act(n, c, h, w, xs[0].device, cs)
        arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        L.call('cat_add_n', arr, len(xs), _p(y), n * h * w * cs, _stream())
        ctx.k = len(xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = conform(dy)
        return tuple(dy for _ in range(ctx.k))


class FanoutFn(torch.autograd.Function):
    """k aliases of one activation; the backward sums the k gradients with one add_n kernel, so autograd never
    accumulates fan-out gradients with its own (layout-breaking) adds."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        ctx.set_materialize_grads(False)      # an unused alias contributes None, not a torch-allocated zero tensor
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *dys):
        dys = [conform(d) for d in dys if d is not None]
        if not dys:
            return None, None
        cur = torch.cuda.current_stream()      # (unconditional: the flag may have changed since the forward pass built this node)
        for d in dys:
            if d.is_cuda:
                d.record_stream(cur)       # produced on a branch stream, consumed (and freed) here
        if len(dys) == 1:
            return dys[0], None
        n, c, h, w = dys[0].shape
        cs = act_cs(dys[0])
        out = empty_act(n, c, h, w, dys[0].device, cs)
        srcs, rest = dys[:8], dys[8:]
        while True:                          # add_n takes up to 8 sources; dst may alias src[0]
            arr = (C.c_void_p * len(srcs))(*[d.data_ptr() for d in srcs])
            L.call('cat_add_n', arr, len(srcs), _p(out), n * h * w * cs, _stream())
            if not rest:
                break
            srcs, rest = [out] + rest[:7], rest[7:]
        return out, None


def fanout(x, k):
    if k == 1:
        return (x,)
    return FanoutFn.apply(x, k)


class Concat2Fn(torch.autograd.Function):
    """torch.cat((a, b), 1) for NHWC activations."""

    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a)
        a, b = conform(a), conform(b)
        n, ca, h, w = a.shape
        cb = b.shape[1]
        y = empty_act(n, ca + cb, h, w, a.device)
        L.call('cat_concat2', _p(a), ca, act_cs(a), _p(b), cb, act_cs(b), _p(y), act_cs(y), n * h * w, _stream())
        ctx.dims = (n, ca, cb, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, ca, cb, h, w = ctx.dims
        dy = conform(dy)
        da = db = None
        if ctx.needs_input_grad[0]:
            da = empty_act(n, ca, h, w, dy.device)
            L.call('cat_slice_channels', _p(dy), act_cs(dy), 0, ca, _p(da), act_cs(da), n * h * w, _stream())
        if ctx.needs_input_grad[1]:
            db = empty_act(n, cb, h, w, dy.device)
            L.call('cat_slice_channels', _p(dy), act_cs(dy), ca, cb, _p(db), act_cs(db), n * h * w, _stream())
        return da, db


# ---------------------------------------------------------------------------------------------- losses
class KAFn(torch.autograd.Function):
    """Kernel alignment between two activation stacks (utils/common.py:38-46); gradient flows to X only."""

    @staticmethod
    def forward(ctx, x, y):
        _require_cuda(x)
        x, y = conform(x), conform(y)
        n = x.shape[0]
        if y.shape[0] != n:
            raise AssertionError(f'X_ and Y_ must have the same shape on dim 0, but got {n} for X_ and {y.shape[0]} for Y_.')
        dx_len = x.shape[2] * x.shape[3] * act_cs(x)
        dy_len = y.shape[2] * y.shape[3] * act_cs(y)
        ws = torch.empty(L.query('cat_ka_ws_bytes', n) // 4, device=x.device, dtype=torch.float32)
        out = torch.empty((), device=x.device, dtype=torch.float32)
        L.call('cat_ka_fwd', _p(x), dx_len, _p(y), dy_len, n, _p(out), _p(ws), _stream())
        ctx.save_for_backward(x, ws)
        ctx.dx_len = dx_len
        return out

    @staticmethod
    def backward(ctx, gout):
        x, ws = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = empty_act(n, c, h, w, x.device, act_cs(x))
        gout = gout.contiguous()
        L.call('cat_ka_bwd', _p(x), ctx.dx_len, n, _p(gout), _p(ws), _p(dx), _stream())
        return dx, None


class LossFn(torch.autograd.Function):
    """mean-reduced scalar losses (L1 / MSE / lsgan / hinge); see cat_loss_fwd in include/cat_hip.h."""

    @staticmethod
    def forward(ctx, a, b, kind, target):
        _require_cuda(a)
        a = conform(a)
        n, c, h, w = a.shape
        if b is not None:
            b = conform(b)
            if b.shape != a.shape or act_cs(b) != act_cs(a):
                raise RuntimeError('loss: operands must share shape and pixel stride')
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = workspace(L.query('cat_loss_ws_bytes', n * h * w), a.device)
        L.call('cat_loss_fwd', kind, _p(a), _p(b), float(target), n * h * w, c, act_cs(a), _p(out), _p(ws), _stream())
        ctx.kind, ctx.target = kind, float(target)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        n, c, h, w = a.shape
        da = empty_act(n, c, h, w, a.device, act_cs(a))
        gout = gout.contiguous()
        L.call('cat_loss_bwd', ctx.kind, _p(a), _p(b), ctx.target, n * h * w, c, act_cs(a), _p(gout), 1.0, _p(da), _stream())
        return da, None, None, None


def ka(x, y):
    return KAFn.apply(x, y)


def fill_(t, value):
    """In-place fill of a dense fp32 buffer with the library's kernel."""
    L.call('cat_fill', _p(t), t.numel(), float(value), _stream())
    return t


# ---------------------------------------------------------------------------------------------- SPADE / GauGAN path
_BN_SYNC = None


def set_bn_sync(sync):
    """`sync`: object with `.world_size` and `.all_reduce_sum_(tensor)` (cat_amd.parallel.DataParallelReducer) or None.  While set,
    training-mode SynchronizedBatchNorm2d layers all-reduce their [sum x | sum x^2] (and backward sums) over the ranks."""
    global _BN_SYNC
    _BN_SYNC = sync if (sync is not None and sync.world_size > 1) else None


def bn_sync():
    return _BN_SYNC


def _bn_forward_stats(x, rm, rv, eps, momentum, gamma, beta, want_affine):
    """Split-phase batch statistics: local sums -> (all-reduce) -> mean / inv_std (+ running stats).  Returns
    (a, b, scale, shift, count, sync): xhat = x*a + b;  y = x*scale + shift (affine map incl. gamma/beta) when want_affine."""
    n, c, h, w = x.shape
    cs = act_cs(x)
    m = n * h * w
    dev = x.device
    st = _stream()
    sync = _BN_SYNC
    sums = torch.empty(2 * cs, device=dev, dtype=torch.float32)
    ws = workspace(L.query('cat_bn_ws_bytes', m, cs), dev)
    L.call('cat_bn_stats_fwd', _p(x), m, c, cs, _p(sums), _p(ws), st)
    count = m
    if sync is not None:
        sync.all_reduce_sum_(sums)
        count = m * sync.world_size
    stats = torch.empty((4 + (2 if want_affine else 0), cs), device=dev, dtype=torch.float32)
    mean, rstd, a, b = stats[0], stats[1], stats[2], stats[3]
    scale, shift = (stats[4], stats[5]) if want_affine else (None, None)
    L.call('cat_bn_finalize', _p(sums), float(count), c, cs, eps, 1 if sync is not None else 0, momentum, _p(gamma), _p(beta), _p(mean),
           _p(rstd), _p(rm), _p(rv), _p(a), _p(b), _p(scale), _p(shift), st)
    return a, b, scale, shift, count, sync


class SyncBNFn(torch.autograd.Function):
    """SynchronizedBatchNorm2d in training mode (+ fused activation): batch statistics over ALL ranks' samples
    (models/modules/sync_batchnorm/batchnorm.py:68-140); with one rank it is F.batch_norm (:69-72)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, eps, momentum, act, slope):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        cs = act_cs(x)
        a, b, scale, shift, count, sync = _bn_forward_stats(x, rm, rv, eps, momentum, gamma, beta, True)
        y = empty_act(n, c, h, w, x.device, cs)
        L.call('cat_affine_act_fwd', _p(x), _p(scale), _p(shift), _p(y), n * h * w, c, cs, act, slope, _stream())
        ctx.meta = (count, sync, act, slope)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.save_for_backward(x, a, b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, b, gamma, beta = ctx.saved_tensors
        count, sync, act, slope = ctx.meta
        n, c, h, w = x.shape
        cs = act_cs(x)
        m = n * h * w
        dy = conform(dy)
        if act_cs(dy) != cs:
            raise RuntimeError('batch norm backward: gradient pixel stride differs from the input')
        st = _stream()
        sums = torch.empty(2 * cs, device=x.device, dtype=torch.float32)
        ws = workspace(L.query('cat_bn_ws_bytes', m, cs), x.device)
        L.call('cat_bn_stats_bwd', _p(x), _p(dy), _p(gamma), _p(beta), _p(a), _p(b), m, c, cs, act, slope, _p(sums), _p(ws), st)
        local = sums
        if sync is not None:
            local = sums.clone()
            sync.all_reduce_sum_(sums)
        need_g = gamma is not None and ctx.needs_input_grad[1]
        need_b = beta is not None and ctx.needs_input_grad[2]
        dx = empty_act(n, c, h, w, x.device, cs) if ctx.needs_input_grad[0] else None
        dgamma = dbeta = None
        tg = _grad_target(ctx.gamma) if need_g else None
        tb = _grad_target(ctx.beta) if need_b else None
        if need_g and need_b and tg is not None and tb is not None:
            sg, sb = ctx.gamma._cat_grad_state, ctx.beta._cat_grad_state
            if sg['fresh'] != sb['fresh']:
                raise RuntimeError('batch norm backward: gamma / beta gradient buffers out of sync')
            acc = 0 if sg['fresh'] else 1
            pg, pb = tg, tb
            sg['fresh'] = sb['fresh'] = False
        else:
            acc = 0
            pg = dgamma = torch.empty_like(gamma) if need_g else None
            pb = dbeta = torch.empty_like(beta) if need_b else None
        L.call('cat_bn_apply_bwd', _p(x), _p(dy), _p(gamma), _p(beta), _p(a), _p(b), _p(sums), float(count), _p(local), _p(dx), _p(pg), _p(pb),
               acc, m, c, cs, act, slope, st)
        return dx, dgamma, dbeta, None, None, None, None, None, None


class SpadeFn(torch.autograd.Function):
    """InceptionSPADE modulation in training mode, fused with the activation behind it:
    y = act(param_free_norm(x) * (1 + gamma) + beta), gamma | beta = channel halves of `gb` (inception_modules.py:746-762)."""

    @staticmethod
    def forward(ctx, x, gb, rm, rv, eps, momentum, act, slope):
        _require_cuda(x)
        x, gb = conform(x), conform(gb)
        n, c, h, w = x.shape
        if gb.shape != (n, 2 * c, h, w):
            raise RuntimeError(f'SPADE: modulation maps have shape {tuple(gb.shape)}, expected {(n, 2 * c, h, w)}')
        cs, gcs = act_cs(x), act_cs(gb)
        a, b, _, _, count, sync = _bn_forward_stats(x, rm, rv, eps, momentum, None, None, False)
        y = empty_act(n, c, h, w, x.device, cs)
        L.call('cat_spade_fwd', _p(x), _p(a), _p(b), _p(gb), _p(y), n * h * w, c, cs, gcs, act, slope, _stream())
        ctx.meta = (count, sync, act, slope)
        ctx.save_for_backward(x, a, b, gb, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, b, gb, y = ctx.saved_tensors
        count, sync, act, slope = ctx.meta
        n, c, h, w = x.shape
        cs, gcs = act_cs(x), act_cs(gb)
        m = n * h * w
        dy = conform(dy)
        if act_cs(dy) != cs:
            raise RuntimeError('SPADE backward: gradient pixel stride differs from the input')
        st = _stream()
        dgb = empty_act(n, 2 * c, h, w, x.device, gcs)
        dx = empty_act(n, c, h, w, x.device, cs)
        sums = torch.empty(2 * cs, device=x.device, dtype=torch.float32)
        ws = workspace(L.query('cat_bn_ws_bytes', m, cs), x.device)
        L.call('cat_spade_bwd_stats', _p(x), _p(a), _p(b), _p(gb), _p(y), _p(dy), _p(dgb), _p(dx), _p(sums), m, c, cs, gcs, act, slope,
               _p(ws), st)
        if sync is not None:
            sync.all_reduce_sum_(sums)
        L.call('cat_spade_bwd_apply', _p(x), _p(a), _p(b), _p(sums), float(count), _p(dx), m, c, cs, st)
        return (dx if ctx.needs_input_grad[0] else None), (dgb if ctx.needs_input_grad[1] else None), None, None, None, None, None, None


class SpadeInstanceFn(torch.autograd.Function):
    """InceptionSPADE modulation whose param-free norm is nn.InstanceNorm2d (norm_G = 'spadeinstance...', reference inception_modules.py:414-415,
    746-762): y = act(instance_norm(x) * (1 + gamma) + beta), in train and eval mode alike (the layer keeps no running statistics).
    Two library calls per pass: cat_norm_fwd in instance mode (statistics about the plane's first pixel -- torch's instance_norm does not suffer
    the sum-of-squares cancellation on near-constant 2 x 4 planes, so neither may this) and the modulation kernels of SpadeFn on the normalised
    tensor (a = 1, b = 0); backward: cat_spade_bwd_stats leaves d xhat in its dx buffer, cat_norm_bwd turns it into dx.  Never all-reduced.
    No launch script uses the option."""

    @staticmethod
    def forward(ctx, x, gb, eps, act, slope):
        _require_cuda(x)
        x, gb = conform(x), conform(gb)
        n, c, h, w = x.shape
        if gb.shape != (n, 2 * c, h, w):
            raise RuntimeError(f'SPADE: modulation maps have shape {tuple(gb.shape)}, expected {(n, 2 * c, h, w)}')
        cs, gcs = act_cs(x), act_cs(gb)
        st = _stream()
        geom = (n, h * w, c, cs, L.NORM_INSTANCE, eps, 0.0, L.ACT_NONE, 0.0)
        g = NormGeom(*geom)
        xh = empty_act(n, c, h, w, x.device, cs)
        mean = torch.empty((n, c), device=x.device, dtype=torch.float32)
        rstd = torch.empty((n, c), device=x.device, dtype=torch.float32)
        ws = workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
        L.call('cat_norm_fwd', C.byref(g), _p(x), None, None, _p(xh), _p(mean), _p(rstd), None, None, None, _p(ws), st)
        ab = torch.zeros((2, cs), device=x.device, dtype=torch.float32)
        ab[0].fill_(1.0)
        y = empty_act(n, c, h, w, x.device, cs)
        L.call('cat_spade_fwd', _p(xh), _p(ab[0]), _p(ab[1]), _p(gb), _p(y), n * h * w, c, cs, gcs, act, slope, st)
        ctx.meta = (geom, act, slope)
        ctx.save_for_backward(x, xh, mean, rstd, ab, gb, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xh, mean, rstd, ab, gb, y = ctx.saved_tensors
        geom, act, slope = ctx.meta
        n, c, h, w = x.shape
        cs, gcs, m = act_cs(x), act_cs(gb), n * h * w
        dy = conform(dy)
        if act_cs(dy) != cs:
            raise RuntimeError('SPADE backward: gradient pixel stride differs from the input')
        st = _stream()
        dgb = empty_act(n, 2 * c, h, w, x.device, gcs)
        dxh = empty_act(n, c, h, w, x.device, cs)
        sums = torch.empty(2 * cs, device=x.device, dtype=torch.float32)
        ws = workspace(L.query('cat_bn_ws_bytes', m, cs), x.device)
        L.call('cat_spade_bwd_stats', _p(xh), _p(ab[0]), _p(ab[1]), _p(gb), _p(y), _p(dy), _p(dgb), _p(dxh), _p(sums), m, c, cs, gcs, act, slope,
               _p(ws), st)
        dx = None
        if ctx.needs_input_grad[0]:
            g = NormGeom(*geom)
            dx = empty_act(n, c, h, w, x.device, cs)
            ws = workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
            L.call('cat_norm_bwd', C.byref(g), _p(x), _p(dxh), None, None, _p(mean), _p(rstd), _p(dx), None, None, 0, _p(ws), st)
        return dx, (dgb if ctx.needs_input_grad[1] else None), None, None, None


def spade_eval(x, gb, rm, rv, eps, act, slope):
    """Frozen (eval, no-grad) SPADE: the param-free norm uses the running statistics."""
    _require_cuda(x)
    x, gb = conform(x), conform(gb)
    if torch.is_grad_enabled() and (x.requires_grad or gb.requires_grad):
        raise NotImplementedError('eval-mode SPADE is only implemented for the frozen (no-grad) teacher')
    n, c, h, w = x.shape
    a, b = bn_fold(None, None, rm, rv, eps)
    cs = act_cs(x)
    if cs != c:     # bn_fold emits C entries; the kernels read cs (zero padded)
        ap = torch.zeros(2, cs, device=x.device, dtype=torch.float32)
        ap[0, :c].copy_(a)
        ap[1, :c].copy_(b)
        a, b = ap[0], ap[1]
    y = empty_act(n, c, h, w, x.device, cs)
    L.call('cat_spade_fwd', _p(x), _p(a), _p(b), _p(gb), _p(y), n * h * w, c, cs, act_cs(gb), act, slope, _stream())
    return y


class InterpNearestFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='nearest') / nn.Upsample(scale_factor=k).  Backward for integer up-scaling only."""

    @staticmethod
    def forward(ctx, x, ho, wo):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        y = empty_act(n, c, ho, wo, x.device)
        L.call('cat_interp_nearest_fwd', _p(x), _p(y), n, h, w, ho, wo, c, act_cs(x), act_cs(y), _stream())
        ctx.dims = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, ho, wo = ctx.dims
        f = ho // h
        if ho != f * h or wo != f * w:
            raise NotImplementedError('nearest interpolation backward: integer up-scaling only')
        dy = conform(dy)
        dx = empty_act(n, c, h, w, dy.device, act_cs(dy))
        L.call('cat_upsample_nearest_bwd', _p(dy), _p(dx), n, h, w, f, c, act_cs(dy), _stream())
        return dx, None, None


def interp_nearest(x, size):
    return InterpNearestFn.apply(x, int(size[0]), int(size[1]))


class AvgPool3x3s2Fn(torch.autograd.Function):
    """F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False) (discriminators.py:213-218)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        y = empty_act(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x.device, act_cs(x))
        L.call('cat_avgpool3x3s2_fwd', _p(x), _p(y), n, h, w, c, act_cs(x), _stream())
        ctx.dims = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.dims
        dy = conform(dy)
        dx = empty_act(n, c, h, w, dy.device, act_cs(dy))
        L.call('cat_avgpool3x3s2_bwd', _p(dy), _p(dx), n, h, w, c, act_cs(dy), _stream())
        return dx


class MaxPool2x2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) (VGG19.features, models/modules/loss.py:151-186)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = conform(x)
        n, c, h, w = x.shape
        y = empty_act(n, c, h // 2, w // 2, x.device, act_cs(x))
        L.call('cat_maxpool2x2_fwd', _p(x), _p(y), n, h, w, c, act_cs(x), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = conform(dy)
        dx = empty_act(n, c, h, w, x.device, act_cs(x))
        L.call('cat_maxpool2x2_bwd', _p(x), _p(dy), _p(dx), n, h, w, c, act_cs(x), _stream())
        return dx


def onehot_edges(label, inst, nc):
    """SPADEModel.preprocess_input (models/spade_model.py:142-179): [N,1,H,W] integer label (+ instance ids) -> input_semantics."""
    _require_cuda(label)
    n, one, h, w = label.shape
    lab = label.to(torch.int32).contiguous()
    ins = None if inst is None else inst.to(torch.int32).contiguous()
    c = nc + (0 if inst is None else 1)
    y = empty_act(n, c, h, w, label.device)
    L.call('cat_onehot_edges', _p(lab), _p(ins), _p(y), n, h, w, nc, act_cs(y), _stream())
    return y


class DiscInputFn(torch.autograd.Function):
    """SPADEModelModules.discriminate's input (spade_model_modules.py:136-141): cat over the batch of [sem | fake] and
    [sem | real].  Gradient flows to `fake` only."""

    @staticmethod
    def forward(ctx, sem, fake, real):
        _require_cuda(sem)
        sem, fake, real = conform(sem), conform(fake), conform(real)
        n, cs_, h, w = sem.shape
        ci = fake.shape[1]
        y = empty_act(2 * n, cs_ + ci, h, w, sem.device)
        ycs = act_cs(y)
        half = n * h * w * ycs * 4
        st = _stream()
        L.call('cat_concat2', _p(sem), cs_, act_cs(sem), _p(fake), ci, act_cs(fake), _p(y), ycs, n * h * w, st)
        L.call('cat_concat2', _p(sem), cs_, act_cs(sem), _p(real), ci, act_cs(real), C.c_void_p(y.data_ptr() + half), ycs, n * h * w, st)
        ctx.dims = (n, cs_, ci, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, cs_, ci, h, w = ctx.dims
        dy = conform(dy)
        dfake = None
        if ctx.needs_input_grad[1]:
            dfake = empty_act(n, ci, h, w, dy.device)
            L.call('cat_slice_channels', _p(dy), act_cs(dy), cs_, ci, _p(dfake), act_cs(dfake), n * h * w, _stream())
        return None, dfake, None


class BatchHalvesFn(torch.autograd.Function):
    """divide_pred (spade_model_modules.py:143-155): the fake / real halves of a tensor computed on the 2N batch."""

    @staticmethod
    def forward(ctx, t):
        t = conform(t)
        n2 = t.shape[0]
        ctx.shape = tuple(t.shape)
        ctx.cs = act_cs(t)
        ctx.set_materialize_grads(False)
        return t[:n2 // 2], t[n2 // 2:]

    @staticmethod
    def backward(ctx, d0, d1):
        if d0 is None and d1 is None:
            return None
        n2, c, h, w = ctx.shape
        dev = (d0 if d0 is not None else d1).device
        dy = empty_act(n2, c, h, w, dev, ctx.cs)
        half = (n2 // 2) * h * w * ctx.cs
        st = _stream()
        for k, d in enumerate((d0, d1)):
            dst = C.c_void_p(dy.data_ptr() + k * half * 4)
            if d is None:
                L.call('cat_fill', dst, half, 0.0, st)
            else:
                d = conform(d)
                arr = (C.c_void_p * 1)(d.data_ptr())
                L.call('cat_add_n', arr, 1, dst, half, st)
        return dy


class SpectralNormFn(torch.autograd.Function):
    """torch.nn.utils.spectral_norm's pre-forward hook (spade_architecture/normalization.py:27-29): one power iteration on the
    persistent u / v (training mode, in place, no grad), weight = weight_orig / sigma."""

    @staticmethod
    def forward(ctx, weight_orig, u, v, power_iter, eps):
        _require_cuda(weight_orig)
        wcl, wcs = weight_cl(weight_orig)
        o, i, kh, kw = weight_orig.shape
        taps = kh * kw
        dev = weight_orig.device
        w_sn = padded_weight_like(weight_orig.shape, dev)
        if weight_wcs(w_sn) != wcs:
            raise RuntimeError('spectral norm: weight layout mismatch')
        vp = torch.empty(taps * wcs, device=dev, dtype=torch.float32)
        sigma = torch.empty(1, device=dev, dtype=torch.float32)
        ws = workspace(L.query('cat_spectral_norm_ws_bytes', o, i, taps, wcs), dev)
        L.call('cat_spectral_norm_fwd', _p(wcl), o, i, taps, wcs, _p(u), _p(v), 1 if power_iter else 0, eps, _p(sigma), _p(w_sn), _p(vp),
               _p(ws), _stream())
        ctx.geom = (o, taps, wcs)
        ctx.weight = weight_orig
        ctx.save_for_backward(w_sn, u.clone() if power_iter else u, vp, sigma)
        return w_sn

    @staticmethod
    def backward(ctx, gw):
        w_sn, u, vp, sigma = ctx.saved_tensors
        o, taps, wcs = ctx.geom
        if weight_wcs(gw) != wcs:
            g2 = padded_weight_like(w_sn.shape, gw.device)
            g2.copy_(gw)
            gw = g2
        ws = workspace(L.query('cat_spectral_norm_ws_bytes', o, 0, taps, wcs), gw.device)
        st = _stream()

        def k(dst, acc):
            if _grad_wcs(dst) != wcs:
                raise RuntimeError('spectral norm backward: gradient buffer layout mismatch')
            L.call('cat_spectral_norm_bwd', _p(gw), _p(w_sn), _p(u), _p(vp), _p(sigma), o, taps, wcs, _p(dst), acc, _p(ws), st)
        dw = _write_param_grad(ctx.weight, k)
        return dw, None, None, None, None
