"""Frozen-teacher fast path for InvertedResidualChannels (reference inception_modules.py:124-180, 230-236).

In the distillation step the teacher runs in eval mode under no_grad (base_inception_distiller.py:168, inception_distiller.py:
100-103).  With BatchNorm (running statistics) every norm of a block is a per-channel affine map, so the block

    x + pw_bn( sum_k res_k(x) + sum_k dw_k(x) )

collapses algebraically -- same values, different evaluation order -- into

    A   one 1x1 conv  x -> [h_res1 | g_dw1 | g_dw3 | g_dw5]   (the four 256->42 convs + BN + ReLU as ONE GEMM with N = 176)
    dw  three depthwise convs (+ BN + ReLU in the epilogue) reading / writing channel SLICES of the concatenated buffers
    F   one 1x1 conv  [h_res1 | h_dw1 | h_dw3 | h_dw5] -> C  (the four 42->256 convs as ONE GEMM with K = 176; pw_bn's scale folded
        into its weights, every bias of the block and pw_bn's shift folded into its bias)
    k>1 res branches keep their first conv (+ BN + ReLU folded) -- with one 5 x 5 and one 3 x 3 branch of the teacher's widths they and A are
        ONE launch that stages x once per 16-channel chunk (cat_tstage1w_fwd); their second convs (pw_bn's scale folded in), F and the skip
    connection become ONE K-concatenated LDS-tile launch (csrc/conv_pk.hip): out = x + [h | hid3 | hid5] * [F ; W3 ; W5] + bias,
    written once -- on planes too small for that kernel: out = add_n(x, F, k3, k5)

instead of 12 convs + 3 activations + add_n(6) + affine + add_n(2): the 42->256 1x1 layers were epilogue-bound (K = 44: 27 TFLOP/s),
and 10 of 18 full-size tensor passes disappear.  One-time weight folding is cached per block (invalidated when a tensor changes).
InstanceNorm teachers (CycleGAN configs) cannot be folded and take the general path."""
import ctypes as C

import torch

from . import _lib as L
from . import nn as cnn
from . import ops
from . import optim


_STAGE1W = True      # stage 1 of the block as one launch (cat_tstage1w_fwd); A/B closed in round 6: teacher forward 11.51 -> 11.15 ms


def _affine(bn):
    scale = torch.rsqrt(bn.running_var + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight
    shift = -bn.running_mean * scale
    if bn.bias is not None:
        shift = shift + bn.bias
    return scale, shift


def _foldable(bn):
    return isinstance(bn, cnn.BatchNorm2d) and bn.track_running_stats and not bn.training


def applicable(block, x):
    if torch.is_grad_enabled() or block.training or not x.is_cuda or len(block.res_ops) + len(block.dw_ops) == 0:
        return False
    if getattr(block, '_cat_frozen_off', False):
        return False
    norms = [op[1][1] for op in block.res_ops] + [op[0][1] for op in block.dw_ops] + [op[2][1] for op in block.dw_ops] + [block.pw_bn]
    if not all(_foldable(n) for n in norms):
        return False
    if block.padding_type not in ('reflect', 'zero'):      # 'replicate' is a materialised pad: general path
        return False
    return block.dropout_rate == 0 and len(block.dw_ops) + sum(1 for op in block.res_ops if op[1][0].kernel_size[0] == 1) >= 2


def _tensors(block):
    out = []
    for m in block.modules():
        for t in list(m._parameters.values()) + list(m._buffers.values()):
            if t is not None:
                out.append(t)
    return out


def _plan(block):
    tensors = _tensors(block)
    # weights owned by a FusedAdam change under torch's feet (raw-pointer kernels): such blocks -- a BatchNorm student run in eval mode by
    # evaluate_model between training steps -- re-fold after every optimizer step; the frozen teacher (no optimizer) folds once
    trainable = any(getattr(t, '_cat_grad_view', None) is not None for t in tensors)
    key = (tuple((t.data_ptr(), t._version) for t in tensors), optim.weights_epoch() if trainable else -1)
    cached = getattr(block, '_cat_frozen', None)
    if cached is not None and cached['key'] == key:
        return cached
    dev = block.pw_bn.running_mean.device
    cin = block.input_dim
    s_pw, t_pw = _affine(block.pw_bn)
    pad_mode = L.PAD_REFLECT if block.padding_type == 'reflect' else L.PAD_ZERO
    slots, wide = [], []          # slots: branches whose LAST conv is 1x1 (res k=1, every dw branch); wide: res branches with k > 1
    for op in block.res_ops:
        (slots if op[1][0].kernel_size[0] == 1 else wide).append(('res', op))
    for op in block.dw_ops:
        slots.append(('dw', op))
    offs, off = [], 0
    for kind, op in slots:
        m = (op[1][0] if kind == 'res' else op[0][0]).out_channels
        offs.append((off, m, ops.cs_for(m)))
        off += ops.cs_for(m)
    hc = off
    w_a = ops.padded_weight_like((hc, cin, 1, 1), dev)
    b_a = torch.zeros(hc, device=dev)
    w_f = ops.padded_weight_like((cin, hc, 1, 1), dev)
    b_f = t_pw.clone()
    dws = []
    act_mod = None
    for (kind, op), (o, m, sz) in zip(slots, offs):
        first = op[1] if kind == 'res' else op[0]             # ConvBNReLU: conv, norm, act
        conv, bn, act_mod = first[0], first[1], first[2]
        s1, t1 = _affine(bn)
        w_a[o:o + m].copy_(conv.weight.detach() * s1.view(-1, 1, 1, 1))
        b_a[o:o + m] = (conv.bias.detach() * s1 if conv.bias is not None else 0) + t1
        last = op[4]
        w_f[:, o:o + m].copy_(last.weight.detach() * s_pw.view(-1, 1, 1, 1))
        if last.bias is not None:
            b_f += last.bias.detach() * s_pw
        if kind == 'dw':
            dconv, dbn = op[2][0], op[2][1]
            s2, t2 = _affine(dbn)
            k = dconv.kernel_size[0]
            dws.append(dict(off=o, m=m, sz=sz, k=k, pad=(k - 1) // 2,
                            w=(dconv.weight.detach() * s2.view(-1, 1, 1, 1)).contiguous(),
                            b=((dconv.bias.detach() * s2 if dconv.bias is not None else 0) + t2).contiguous()))
    wides = []
    for kind, op in wide:
        last = op[4]
        w2 = ops.padded_weight_like(tuple(last.weight.shape), dev)
        w2.copy_(last.weight.detach() * s_pw.view(-1, 1, 1, 1))
        if last.bias is not None:
            b_f += last.bias.detach() * s_pw
        fconv, fbn = op[1][0], op[1][1]
        wf, bf = cnn._folded_eval_bn(fconv, fbn, 0)
        wides.append(dict(op=op, w2=w2, k=last.kernel_size[0], m=fconv.out_channels, wf=wf, bf=bf, act=cnn._act_code(op[1][2])))
    act, slope = cnn._act_code(act_mod)
    # all depthwise convs (+ the copy of the k = 1 residual branch's hidden slice) as ONE launch (cat_dwconv2d_multi_fwd): filters in a 5 x 5
    # frame [25][hc], folded bias, kernel size per channel quad.  The copy is k = 1 with centre weight 1: act(h) = h for an activation
    # that is idempotent (ReLU; LeakyReLU is not: slope^2 on negative values) -- anything else keeps the separate launches
    dwm = None
    if dws and hc // 4 <= L.DWMULTI_MAXQ and all(d['k'] in (1, 3, 5) for d in dws) and act in (L.ACT_RELU, L.ACT_NONE):
        frame = torch.zeros((25, hc), device=dev)
        fbias = torch.zeros(hc, device=dev)
        ks = [1] * (hc // 4)
        for (kind, _), (o, m, sz) in zip(slots, offs):
            if kind == 'res':
                frame[12, o:o + m] = 1.0
        for d in dws:
            k, o2 = d['k'], 2 - d['k'] // 2
            for ky in range(k):
                for kx in range(k):
                    frame[(o2 + ky) * 5 + o2 + kx, d['off']:d['off'] + d['m']] = d['w'][:, 0, ky, kx]
            fbias[d['off']:d['off'] + d['m']] = d['b']
            for q in range(d['off'] // 4, (d['off'] + d['sz']) // 4):
                ks[q] = k
        dwm = dict(frame=frame.contiguous(), bias=fbias.contiguous(), ks=ks)
    # stage 1 in one launch (cat_tstage1w_fwd): one 5 x 5 and one 3 x 3 residual branch + the concatenated 1 x 1 convs, the teacher's widths
    s1w = None
    by_k = {wd['k']: wd for wd in wides}
    if (_STAGE1W and len(wides) == 2 and set(by_k) == {3, 5} and all(wd['act'] == (act, slope) for wd in wides)
            and L.query('cat_tstage1w_supported', by_k[5]['m'], by_k[3]['m'], hc)):
        s1w = dict(packs=None, ws=[by_k[5]['wf'], by_k[3]['wf'], w_a],      # packed filter streams: built at the first launch
                   biases=[by_k[5]['bf'], by_k[3]['bf'], b_a.contiguous()], m=[by_k[5]['m'], by_k[3]['m'], hc])
    plan = dict(key=key, s1w=s1w, tail_pack=None, tail_offs=None, hc=hc, dwm=dwm, w_a=w_a, b_a=b_a.contiguous(), w_f=w_f, b_f=b_f.contiguous(), dws=dws, wides=wides, act=act, slope=slope,
                pad_mode=pad_mode, copies=[(o, sz) for (kind, _), (o, m, sz) in zip(slots, offs) if kind == 'res'])
    block._cat_frozen = plan
    return plan


def block_forward(block, x):
    p = _plan(block)
    x = ops.conform(x)
    n, c, h, w = x.shape
    hc = p['hc']
    m_pix = n * h * w

    def dw_stage(hbuf):
        # every depthwise conv (+ folded BN + activation) and the k = 1 residual branch's copy as one launch
        h2 = ops.empty_act(n, hc, h, w, hbuf.device)
        gm = L.DwMulti()
        gm.N, gm.H, gm.W, gm.nq, gm.xcs, gm.ycs = n, h, w, hc // 4, hc, hc
        gm.reflect, gm.act, gm.slope = int(p['pad_mode'] == L.PAD_REFLECT), p['act'], p['slope']
        for q, k in enumerate(p['dwm']['ks']):
            gm.ks[q] = k
        L.call('cat_dwconv2d_multi_fwd', C.byref(gm), ops._p(hbuf), ops._p(p['dwm']['frame']), ops._p(p['dwm']['bias']), ops._p(h2), ops._stream())
        return h2

    def concat_chain(xi):
        # A: every first-level 1x1 conv (+ folded BN + activation) as one GEMM
        stc = ops._stream()
        from . import ksum
        aseg = [ksum.Segment(xi, p['w_a'], False)]
        if ksum.applicable(aseg, n, h, w, hc):      # 176 wide: two 96-column tiles of the DMA-staged GEMM (8 % padding; 128-wide tiles: 31 %)
            hbuf = ksum.run(aseg, p['b_a'], ops.empty_act(n, hc, h, w, xi.device), act=p['act'], slope=p['slope'])
        else:
            hbuf = ops.Conv2dFn.apply(xi, p['w_a'], p['b_a'], 1, 0, L.PAD_ZERO, p['act'], p['slope'])
        if p['dwm'] is not None:
            h2 = dw_stage(hbuf)
            if fused_tail:
                return h2
            return ops.Conv2dFn.apply(h2, p['w_f'], p['b_f'], 1, 0, L.PAD_ZERO, L.ACT_NONE, 0.0)
        h2 = ops.empty_act(n, hc, h, w, xi.device)
        for o, sz in p['copies']:       # k = 1 res branch: its hidden activation already is the last conv's input
            L.call('cat_slice_channels', ops._p(hbuf), hc, o, sz, C.c_void_p(h2.data_ptr() + 4 * o), hc, m_pix, stc)
        for d in p['dws']:              # depthwise k x k (+ folded BN + activation) on a channel slice of the concatenated buffers
            g = ops._conv_geom(n, h, w, d['m'], hc, h, w, d['m'], hc, d['k'], d['k'], 1, d['pad'], p['pad_mode'], p['act'], p['slope'], d['sz'])
            L.call('cat_dwconv2d_fwd', C.byref(g), C.c_void_p(hbuf.data_ptr() + 4 * d['off']), ops._p(d['w']), ops._p(d['b']),
                   C.c_void_p(h2.data_ptr() + 4 * d['off']), stc)
        if fused_tail:
            return h2
        # F: every last 1x1 conv as one GEMM (pw_bn scale in the weights; all biases + pw_bn shift in the bias)
        return ops.Conv2dFn.apply(h2, p['w_f'], p['b_f'], 1, 0, L.PAD_ZERO, L.ACT_NONE, 0.0)

    def wide_chain(wd):
        def run(xi):
            op = wd['op']
            hid = op[1](op[0](xi))      # pad -> conv k x k with folded BN + activation (FusedSequential's frozen path)
            if fused_tail:
                return hid
            return ops.Conv2dFn.apply(hid, wd['w2'], None, 1, (wd['k'] - 1) // 2, p['pad_mode'], L.ACT_NONE, 0.0)
        return run

    fused_tail = ops.tconv_applicable(n, h, w, c, 3, 3, 1, 1) and 1 + len(p['wides']) <= L.TCONV_MAXSEG
    fns = [concat_chain] + [wide_chain(wd) for wd in p['wides']]
    s1w = p['s1w'] if fused_tail and p['dwm'] is not None else None
    if s1w is not None:
        # every first conv of the block from one staging of x: hid5, hid3 and the concatenated 1 x 1 hidden buffer in one launch
        if s1w['packs'] is None:
            from . import tconv
            s1w['packs'] = [tconv.pack(wt, tconv.FWD) for wt in s1w['ws']]
        g = L.Stage1WGeom()
        g.N, g.H, g.W, g.xcs, g.cin = n, h, w, ops.act_cs(x), c
        g.reflect, g.act, g.slope = int(p['pad_mode'] == L.PAD_REFLECT), p['act'], p['slope']
        bufs = [ops.empty_act(n, m, h, w, x.device) for m in s1w['m']]
        for k in range(3):
            g.ycs[k], g.nvalid[k] = ops.act_cs(bufs[k]), s1w['m'][k]
        arr = C.c_void_p * 3
        L.call('cat_tstage1w_fwd', C.byref(g), ops._p(x), arr(*[t.data_ptr() for t in s1w['packs']]), arr(*[t.data_ptr() for t in s1w['biases']]),
               arr(*[t.data_ptr() for t in bufs]), ops._stream())
        hid = {5: bufs[0], 3: bufs[1]}
        outs = [dw_stage(bufs[2])] + [hid[wd['k']] for wd in p['wides']]
    elif ops.branch_streams_enabled() and len(fns) > 1:
        outs = ops.run_on_side_streams(fns, [x] * len(fns))
    else:
        outs = [fn(x) for fn in fns]
    if not fused_tail:
        return ops.AddNFn.apply(x, *outs)
    from . import ksum, tconv
    refl = p['pad_mode'] == L.PAD_REFLECT
    # wide output (the teacher's C = 256): the tail as ONE implicit GEMM on 128 x 128 tiles with both operands DMA-ed into LDS
    ksegs = [ksum.Segment(outs[0], p['w_f'], False)] + [ksum.Segment(hid, wd['w2'], refl) for wd, hid in zip(p['wides'], outs[1:])]
    if ksum.applicable(ksegs, n, h, w, c):
        return ksum.run(ksegs, p['b_f'], ops.empty_act(n, c, h, w, x.device), res=x)
    if p['tail_pack'] is None:      # packed filters of the fused tail (once per plan): segment 0 = F, then one segment per wide branch
        packs = [tconv.pack(p['w_f'], tconv.FWD)] + [tconv.pack(wd['w2'], tconv.FWD) for wd in p['wides']]
        offs, po = [], 0
        for pk in packs:
            offs.append(po)
            po += pk.numel()
        p['tail_pack'], p['tail_offs'] = torch.cat(packs), offs
    segs = [tconv.Segment(outs[0], 1, 0, False, p['tail_offs'][0])]
    for wd, hid, off in zip(p['wides'], outs[1:], p['tail_offs'][1:]):
        segs.append(tconv.Segment(hid, wd['k'], (wd['k'] - 1) // 2, refl, off))
    y = ops.empty_act(n, c, h, w, x.device)
    return tconv.run(segs, p['tail_pack'], p['b_f'], y, c, n, h, w, h, w, res=x)
