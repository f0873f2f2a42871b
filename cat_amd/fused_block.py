"""Fused training-mode path of InvertedResidualChannels (reference models/modules/inception_modules.py:124-180, 230-236):

    x + pw_bn( sum_k conv2_k(relu(norm(conv1_k(x)))) + sum_k pw2_k(relu(norm(dw_k(relu(norm(pw1_k(x))))))) )

with train-mode BatchNorm2d / InstanceNorm2d in every position.  The general path launches ~55 kernels per block (12 convs, 10 norms
x 3, add_n, ...), most of them a few microseconds of HBM-bound work between dependent launches.  Here a block is 9 launches:

    stage 1   one LDS-tile conv launch per first-conv kernel size (the 1x1 convs of all branches as ONE N-concatenated GEMM) writing
              channel slices of one pre-norm buffer Z1 and the per-tile statistics of every branch           (cat_tconv_fwd + stats)
    finalize  scale / shift of all stage-1 norms + their running statistics                                   (cat_tnorm_finalize)
    dw        all depthwise convs as one launch; normalise + ReLU of stage 1 applied while staging            (cat_dwm_fwd)
    finalize
    stage 2   the branch sum: six second convs K-concatenated into one launch, normalise + ReLU applied while staging, output
              written once, statistics for pw_bn                                                              (cat_tconv_fwd)
    finalize, out = x + T * scale + shift                                                                     (cat_affine_res_fwd)

Filter streams, concatenated gamma / beta / bias vectors and the depthwise filter frame are refreshed by ONE table-driven launch per
block and optimizer step (cat_prep_run).  The backward pass re-materialises the two hidden activations (one pass each) and then runs
the branch-wise weight-gradient kernels on channel slices; the six first-conv input gradients are one K-concatenated launch.
Same arithmetic as the general path (same conv accumulation order per output, statistics merged pairwise instead of sequentially):
results agree to ~1e-6 relative; tests/test_fused_block_gpu.py pins both against the CPU oracle."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import nn as cnn
from . import ops
from . import optim
from . import tconv

_ENABLED = os.environ.get('CAT_FUSED_BLOCK', '1') != '0'
_MERGE_DW_DGRAD = True      # one N-concatenated launch for the depthwise branches' second-conv input gradients (A/B closed in round 4)
_BACKWARD_READY = True


def set_enabled(on):
    global _ENABLED
    _ENABLED = bool(on)


def _cs4(c):
    return (c + 3) // 4 * 4


def _has_hooks(mod):
    for m in mod.modules():
        if m is mod:
            continue
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            return True
    return False


def applicable(block, x):
    if not _ENABLED or not block.training or not x.is_cuda or block.dropout_rate != 0 or block.padding_type not in ('reflect', 'zero'):
        return False
    if torch.is_grad_enabled() and not _BACKWARD_READY:
        return False
    if len(block.res_ops) + len(block.dw_ops) == 0 or not ops.is_act(x):
        return False
    n, c, h, w = x.shape
    if not ops.tconv_applicable(n, h, w, c, 3, 3, 1, 1):
        return False
    norms = [op[1][1] for op in block.res_ops] + [op[0][1] for op in block.dw_ops] + [op[2][1] for op in block.dw_ops] + [block.pw_bn]
    kind = type(norms[0])
    if kind not in (cnn.BatchNorm2d, cnn.InstanceNorm2d) or any(type(m) is not kind for m in norms):
        return False
    if kind is cnn.BatchNorm2d and any(m.momentum is None or not m.track_running_stats for m in norms):
        return False
    if kind is cnn.InstanceNorm2d and any(m.track_running_stats for m in norms):
        return False
    first_act = block.res_ops[0][1][2] if len(block.res_ops) else block.dw_ops[0][0][2]
    if cnn._act_code(first_act)[0] not in (L.ACT_RELU, L.ACT_LRELU):     # the staging paths apply ReLU / LeakyReLU only (nn.ReLU6: general path)
        return False
    ks = [op[1][0].kernel_size[0] for op in block.res_ops] + [op[2][0].kernel_size[0] for op in block.dw_ops]
    if any(k not in (1, 3, 5) for k in ks):
        return False
    if sum(_cs4(op[0][0].out_channels) for op in block.dw_ops) > 4 * L.DWM_MAXQ_BWD or len(block.res_ops) + len(block.dw_ops) > L.TCONV_MAXSEG:
        return False
    return not _has_hooks(block)


class _Plan:
    """Static layout of one block: channel slices, persistent operand buffers and the preparation job table."""

    def __init__(self, block, dev):
        self.block = block
        self.dev = dev
        for m in block.modules():       # conv weights into the kernels' [O][kh][kw][round_up(I, 4)] storage (as Conv2d.forward does lazily)
            if isinstance(m, cnn.Conv2d) and m.groups == 1:
                cnn._to_channels_last_(m)
        C_ = block.input_dim
        self.C, self.cs = C_, _cs4(C_)
        self.reflect = block.padding_type == 'reflect'
        first_act = (block.res_ops[0][1][2] if len(block.res_ops) else block.dw_ops[0][0][2])
        self.act, self.slope = cnn._act_code(first_act)
        self.instance = isinstance(block.pw_bn, cnn.InstanceNorm2d)
        pw = block.pw_bn
        self.eps, self.momentum = float(pw.eps), float(pw.momentum if pw.momentum is not None else 0.0)
        # branches; stage-1 channel order: [res k=1 | dw ... | res k=3 | res k=5] so that same-kernel first convs are adjacent (N concat)
        # and the depthwise inputs are one contiguous slice range
        res = [dict(kind='res', k=op[1][0].kernel_size[0], m=op[1][0].out_channels, conv1=op[1][0], bn1=op[1][1], conv2=op[4]) for op in block.res_ops]
        dws = [dict(kind='dw', k=1, kd=op[2][0].kernel_size[0], m=op[0][0].out_channels, conv1=op[0][0], bn1=op[0][1], dconv=op[2][0], bn2=op[2][1],
                    conv2=op[4]) for op in block.dw_ops]
        order = [b for b in res if b['k'] == 1] + dws + [b for b in res if b['k'] == 3] + [b for b in res if b['k'] == 5]
        off = 0
        for b in order:
            b['o1'], b['w1'] = off, _cs4(b['m'])
            off += b['w1']
        self.hc1 = off
        off = 0
        for b in dws:
            b['od'] = off
            off += _cs4(b['m'])
        self.hcd = off
        self.dw_in0 = dws[0]['o1'] if dws else 0
        self.branches, self.res, self.dws = order, res, dws
        # stage-1 launches: one per first-conv kernel size
        self.groups = []
        for k in (1, 3, 5):
            bs = [b for b in order if b['k'] == k]
            if bs:
                g0, g1 = bs[0]['o1'], bs[-1]['o1'] + bs[-1]['w1']
                self.groups.append(dict(k=k, off=g0, width=g1 - g0, branches=bs))
        z = lambda n: torch.zeros(max(n, 4), device=dev, dtype=torch.float32)
        cin4 = self.cs
        # persistent operands
        for g in self.groups:
            g['pack'] = z(tconv.pack_floats(g['k'], cin4, g['width']))
            g['dpack'] = None
        self.gamma1, self.beta1, self.bias1 = z(self.hc1), z(self.hc1), z(self.hc1)
        self.gammad, self.betad, self.biasd = z(self.hcd), z(self.hcd), z(self.hcd)
        self.bias2 = z(self.cs)
        self.w25 = z(25 * self.hcd)
        self.has_bias1 = any(b['conv1'].bias is not None for b in order)
        self.has_biasd = any(b['dconv'].bias is not None for b in dws)
        self.has_bias2 = any(b['conv2'].bias is not None for b in order)
        self.affine = pw.weight is not None
        # stage-2 (branch sum) filter stream: one segment per branch
        po = 0
        for b in order:
            k2 = b['k'] if b['kind'] == 'res' else 1
            b['k2'], b['p2off'] = k2, po
            po += tconv.pack_floats(k2, b['w1'], self.C)
        self.pack2 = z(po)
        # backward filter streams: input gradients of the second convs (per branch) and of the first convs (K-concatenated)
        po = 0
        for b in order:
            b['d2off'] = po
            po += tconv.pack_floats(b['k2'], self.cs, b['m'])
        self.dpack2 = z(po)
        # ... the 1 x 1 second convs of the depthwise branches N-concatenated: their input gradients are ONE launch over dT into dAd
        self.dpack2_dw = z(tconv.pack_floats(1, self.cs, self.hcd)) if (_MERGE_DW_DGRAD and len(dws) > 1) else None
        po = 0
        for b in order:
            b['d1off'] = po
            po += tconv.pack_floats(b['k'], b['w1'], self.C)
        self.dpack1 = z(po)
        # concatenated parameter gradients (norm gamma / beta, conv biases) and where their slices go
        self.gv = dict(g1=z(self.hc1), b1=z(self.hc1), c1=z(self.hc1), gd=z(self.hcd), bd=z(self.hcd), cd=z(self.hcd), c2=z(self.cs))
        self.targets = []       # (vector name, offset, n, parameter)
        for b in order:
            if b['bn1'].weight is not None:
                self.targets += [('g1', b['o1'], b['m'], b['bn1'].weight), ('b1', b['o1'], b['m'], b['bn1'].bias)]
            if b['conv1'].bias is not None:
                self.targets.append(('c1', b['o1'], b['m'], b['conv1'].bias))
            if b['conv2'].bias is not None:
                self.targets.append(('c2', 0, self.C, b['conv2'].bias))
        for b in dws:
            if b['bn2'].weight is not None:
                self.targets += [('gd', b['od'], b['m'], b['bn2'].weight), ('bd', b['od'], b['m'], b['bn2'].bias)]
            if b['dconv'].bias is not None:
                self.targets.append(('cd', b['od'], b['m'], b['dconv'].bias))
        # merged weight-gradient launches: the 1 x 1 first convs of all branches are one GEMM (N-concatenated, rows of `w1` then go to
        # the four parameters), the 1 x 1 second convs of the depthwise branches one K-concatenated GEMM (columns of `w2`)
        g1 = next((g for g in self.groups if g['k'] == 1), None)
        self.merge1 = g1 if (g1 is not None and len(g1['branches']) > 1) else None
        if self.merge1 is not None:
            self.gv['w1'] = z(g1['width'] * self.cs)
            for b in g1['branches']:
                self.targets.append(('w1', (b['o1'] - g1['off']) * self.cs, b['m'] * self.cs, b['conv1'].weight))
        self.merge2 = len(dws) > 1
        self.targets2d = []      # (vector, src offset, rows, cols, src stride, parameter): dst stride = the parameter's own wcs
        if self.merge2:
            self.gv['w2'] = z(self.C * self.hcd)
            for b in dws:
                self.targets2d.append(('w2', b['od'], self.C, _cs4(b['m']), self.hcd, b['conv2'].weight))
        self.scatter_jobs = None
        self._build_jobs()
        self.key = None
        self.bkey = None

    def __deepcopy__(self, memo):
        """A copied module builds its own plan at its first forward (plans hold device buffers, job tables with raw parameter addresses and
        the group of blocks they are prepared with: none of that belongs to the copy)."""
        return None

    # -- job tables ------------------------------------------------------------------------------------------------------
    def _jobs_to_dev(self, jobs):
        arr = (L.PrepJob * len(jobs))()
        blk = 0
        for i, j in enumerate(jobs):
            for f, v in j.items():
                if f == 'srcs':
                    for k, pv in enumerate(v):
                        arr[i].srcs[k] = pv
                elif f != 'threads':
                    setattr(arr[i], f, v)
            nb = max(1, (j['threads'] + 255) // 256)
            arr[i].block0, arr[i].nblocks = blk, nb
            blk += nb
        raw = bytes(arr)
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        self._last_arr = arr      # host copy (prepare_many merges the tables of several blocks into one launch)
        return t, len(jobs), blk

    def _pack_job(self, w, dst_ptr, mode, nn, ck, ks, nt_total, col0):
        wcl, wcs = ops.weight_cl(w)
        if wcl.data_ptr() != w.data_ptr():
            raise RuntimeError('fused block: conv weights must be in kernel layout')
        taps = ks * ks
        c4 = _cs4(ck)
        nfull, rem = c4 // 16, (c4 % 16) // 4
        groups = nfull * taps + ((taps * rem + 3) // 4 if rem else 0)
        ntw = (col0 + nn + 15) // 16 - col0 // 16
        return dict(kind=0, srcs=[w.data_ptr()], dst=dst_ptr, mode=mode, Nn=nn, Ck=ck, ks=ks, wcs=wcs, wn=taps * wcs, c4=c4, nt_total=nt_total, col0=col0,
                    threads=groups * ntw * 64)

    def _build_jobs(self):
        fwd, bwd = [], []
        vec = lambda dst, off, srcs, n: dict(kind=1, srcs=[s.data_ptr() for s in srcs], nsrc=len(srcs), dst=dst.data_ptr() + 4 * off, n=n, threads=n)
        for g in self.groups:
            nt = (g['width'] + 15) // 16
            for b in g['branches']:
                fwd.append(self._pack_job(b['conv1'].weight, g['pack'].data_ptr(), tconv.FWD, b['m'], self.C, g['k'], nt, b['o1'] - g['off']))
        for b in self.branches:
            if b['bn1'].weight is not None:
                fwd.append(vec(self.gamma1, b['o1'], [b['bn1'].weight], b['m']))
                fwd.append(vec(self.beta1, b['o1'], [b['bn1'].bias], b['m']))
            if b['conv1'].bias is not None:
                fwd.append(vec(self.bias1, b['o1'], [b['conv1'].bias], b['m']))
            nt2 = (self.C + 15) // 16
            fwd.append(self._pack_job(b['conv2'].weight, self.pack2.data_ptr() + 4 * b['p2off'], tconv.FWD, self.C, b['m'], b['k2'], nt2, 0))
            bwd.append(self._pack_job(b['conv2'].weight, self.dpack2.data_ptr() + 4 * b['d2off'], tconv.DGRAD, b['m'], self.C, b['k2'], (b['m'] + 15) // 16, 0))
            bwd.append(self._pack_job(b['conv1'].weight, self.dpack1.data_ptr() + 4 * b['d1off'], tconv.DGRAD, self.C, b['m'], b['k'], nt2, 0))
            if self.dpack2_dw is not None and b['kind'] == 'dw':
                bwd.append(self._pack_job(b['conv2'].weight, self.dpack2_dw.data_ptr(), tconv.DGRAD, b['m'], self.C, 1, (self.hcd + 15) // 16, b['od']))
        for b in self.dws:
            if b['bn2'].weight is not None:
                fwd.append(vec(self.gammad, b['od'], [b['bn2'].weight], b['m']))
                fwd.append(vec(self.betad, b['od'], [b['bn2'].bias], b['m']))
            if b['dconv'].bias is not None:
                fwd.append(vec(self.biasd, b['od'], [b['dconv'].bias], b['m']))
            kd = b['kd']
            wd = b['dconv'].weight
            if not wd.is_contiguous():
                raise RuntimeError('fused block: depthwise weights must be contiguous')
            fwd.append(dict(kind=2, srcs=[wd.data_ptr()], dst=self.w25.data_ptr(), Nn=b['m'], ks=kd, col0=b['od'], cs=self.hcd, threads=b['m'] * kd * kd))
        b2 = [b['conv2'].bias for b in self.branches if b['conv2'].bias is not None]
        if b2:
            fwd.append(vec(self.bias2, 0, b2, self.C))
        self.ptrs = tuple(p.data_ptr() for p in self.block.parameters())
        self.shapes = tuple(tuple(p.shape) for p in self.block.parameters())
        self.ids = tuple(id(p) for p in self.block.parameters())
        self.fwd_jobs = self._jobs_to_dev(fwd)
        self.fwd_arr = self._last_arr
        self.bwd_jobs = self._jobs_to_dev(bwd)
        self.bwd_arr = self._last_arr
        self.tables_version = getattr(self, 'tables_version', 0) + 1

    def _epoch_key(self):
        trainable = any(getattr(p, '_cat_grad_view', None) is not None for p in self.block.parameters())
        return (optim.weights_epoch() if trainable else -1, tuple(p._version for p in self.block.parameters()))

    def ptrs_now(self):
        return tuple(q.data_ptr() for q in self.block.parameters())

    def prepare(self, backward=False):
        """Refresh the derived operands if a weight changed since the last refresh (once per optimizer step)."""
        group = getattr(self, 'group', None)
        if backward and group is not None:
            prepare_many(group, backward=True)      # the first backward of the step refreshes every block of the generator in ONE launch
        if tuple(q.data_ptr() for q in self.block.parameters()) != self.ptrs:
            # parameter storage moved since the tables were built (FusedAdam flattens its parameters at its first zero_grad / step,
            # i.e. between the first forward and the first backward): same layout, new source addresses
            self._build_jobs()
            self.key = self.bkey = self.scatter_jobs = None
        key = self._epoch_key()
        if backward:
            if self.bkey != key:
                t, n, blocks = self.bwd_jobs
                L.call('cat_prep_run', ops._p(t), n, blocks, 0, ops._stream())
                self.bkey = key
        elif self.key != key:
            t, n, blocks = self.fwd_jobs
            L.call('cat_prep_run', ops._p(t), n, blocks, 0, ops._stream())
            self.key = key


_MERGE_PREP = True      # one table-driven preparation launch per generator and step (A/B closed in round 4)


def prepare_many(blocks, backward=False):
    """The per-step operand preparation (filter packing, parameter gathers) of ALL fused blocks of a generator as ONE table-driven launch
    instead of one ~12 us launch per block (9 + 9 per step; 0.11 ms of the 2.97 ms student forward).  Blocks without a plan yet (first
    forward) or whose operands are current are left to their own prepare()."""
    if not _MERGE_PREP:
        return
    plans = [getattr(b, '_cat_fused_plan', None) for b in blocks]
    prepare_plans([p for p in plans if p is not None], blocks, backward)


def prepare_plans(plans, group, backward=False):
    """prepare_many over plan objects (fused_block._Plan / fused_spade._Plan: same table fields)."""
    if not _MERGE_PREP:
        return
    stale = []
    for p in plans:
        if p.ptrs_now() != p.ptrs:
            p._build_jobs()
            p.key = p.bkey = p.scatter_jobs = None
        key = p._epoch_key()
        if (p.bkey if backward else p.key) != key:
            stale.append((p, key))
        p.group = group
    if len(stale) < 2:
        return
    # the merged table lives on the first plan and holds the plans it was built from (their ids stay unique while it exists)
    sig = (tuple(id(p) for p, _ in stale), tuple(p.tables_version for p, _ in stale))
    cache = stale[0][0].__dict__.setdefault('_merged', {})
    ent = cache.get(backward)
    ent = ent[1] if ent is not None and ent[0] == sig else None
    if ent is None:
        arrs = [(p.bwd_arr if backward else p.fwd_arr) for p, _ in stale]
        total = sum(len(a) for a in arrs)
        merged = (L.PrepJob * total)()
        i, blk = 0, 0
        for a in arrs:
            for j in a:
                C.memmove(C.byref(merged[i]), C.byref(j), C.sizeof(L.PrepJob))
                merged[i].block0 = blk
                blk += j.nblocks
                i += 1
        t = torch.frombuffer(bytearray(bytes(merged)), dtype=torch.uint8).to(stale[0][0].dev)
        ent = (t, total, blk, tuple(p for p, _ in stale))
        cache[backward] = (sig, ent)
    t, n, nblk = ent[:3]
    L.call('cat_prep_run', ops._p(t), n, nblk, 0, ops._stream())
    for p, key in stale:
        if backward:
            p.bkey = key
        else:
            p.key = key


def plan_for(block, x):
    p = getattr(block, '_cat_fused_plan', None)
    # the plan holds Parameter OBJECTS (gradient targets are keyed by identity): a parameter replaced by one of the same shape
    # (module.weight = nn.Parameter(...), weight transfer) invalidates it like a shape change does
    if p is None or p.dev != x.device or p.shapes != tuple(tuple(q.shape) for q in block.parameters()) or \
            p.ids != tuple(id(q) for q in block.parameters()):
        p = _Plan(block, x.device)
        block._cat_fused_plan = p
    return p


def _slices(pairs):
    arr = (L.NSlice * len(pairs))()
    for i, (c0, c, bn) in enumerate(pairs):
        arr[i].c0, arr[i].c = c0, c
        track = isinstance(bn, cnn.BatchNorm2d) and bn.training and bn.track_running_stats
        arr[i].running_mean = bn.running_mean.data_ptr() if track else None
        arr[i].running_var = bn.running_var.data_ptr() if track else None
        arr[i].num_batches = bn.num_batches_tracked.data_ptr() if track else None
    return arr


def _finalize(p, part, scs, n, h, w, gamma, beta, pairs, mstride=None):
    """-> (ss, mr): ss[0] / ss[1] = scale / shift [G][scs]; mr[0] / mr[1] = mean / rstd [G][mstride] (kept for the backward pass)."""
    G = n if p.instance else 1
    mstride = scs if mstride is None else mstride
    ss = torch.empty((2, G, scs), device=part.device, dtype=torch.float32)
    mr = torch.empty((2, G, mstride), device=part.device, dtype=torch.float32)
    sl = _slices(pairs)
    L.call('cat_tnorm_finalize', ops._p(part), scs, G, n, h, w, ops._p(gamma) if p.affine else None, ops._p(beta) if p.affine else None, len(pairs), sl,
           p.eps, p.momentum, ops._p(ss[0]), ops._p(ss[1]), ops._p(mr[0]), ops._p(mr[1]), mstride, ops._stream())
    return ss, mr


def forward(block, x, save=None):
    """The block's forward on the fused path.  `save`: dict that receives what the backward pass needs (None under no_grad)."""
    p = plan_for(block, x)
    p.prepare()
    n, c, h, w = x.shape
    dev = x.device
    tiles = n * ((h + 7) // 8) * ((w + 15) // 16)
    sstride_of = lambda scs: scs if p.instance else 0
    # ---- stage 1: first convs -> Z1 (pre-norm, concatenated) + tile statistics
    z1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
    part1 = torch.empty((tiles, 2, p.hc1), device=dev, dtype=torch.float32)
    by_k = {g['k']: g for g in p.groups}
    if len(p.groups) == 3 and L.query('cat_tstage1_supported', by_k[5]['width'], by_k[3]['width'], by_k[1]['width']):
        # one launch: the three kernel sizes share every staged input tile
        gs = L.Stage1Geom()
        gs.N, gs.H, gs.W, gs.xcs, gs.cin, gs.reflect, gs.ycs, gs.scs = n, h, w, ops.act_cs(x), c, int(p.reflect), p.hc1, p.hc1
        packs = (C.c_void_p * 3)()
        for slot, k in enumerate((5, 3, 1)):
            g = by_k[k]
            gs.col0[slot], gs.width[slot], gs.nvalid[slot] = g['off'], g['width'], sum(b['m'] for b in g['branches'])
            packs[slot] = g['pack'].data_ptr()
        L.call('cat_tstage1_fwd', C.byref(gs), ops._p(x), packs, ops._p(p.bias1) if p.has_bias1 else None, ops._p(z1), ops._p(part1), ops._stream())
    else:
        for g in p.groups:
            pad = (g['k'] - 1) // 2
            seg = tconv.Segment(x, g['k'], pad, p.reflect and pad > 0, 0)
            tconv.run([seg], g['pack'], (p.bias1.data_ptr() + 4 * g['off']) if p.has_bias1 else None, None, g['width'], n, h, w, h, w, ycs=p.hc1,
                      ycw=g['width'], yptr=z1.data_ptr() + 4 * g['off'], stats=part1.data_ptr() + 4 * g['off'], scs=p.hc1,
                      nvalid=sum(b['m'] for b in g['branches']))
    st1 = _finalize(p, part1, p.hc1, n, h, w, p.gamma1, p.beta1, [(b['o1'], b['m'], b['bn1']) for b in p.branches])
    # ---- depthwise stage
    zd = std = None
    if p.dws:
        zd = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
        partd = torch.empty((tiles, 2, p.hcd), device=dev, dtype=torch.float32)
        gd = L.DwmGeom()
        gd.N, gd.H, gd.W, gd.nq, gd.xcs, gd.ycs, gd.scs = n, h, w, p.hcd // 4, p.hc1, p.hcd, p.hcd
        gd.sstride, gd.reflect, gd.act, gd.slope = sstride_of(p.hc1), int(p.reflect), p.act, p.slope
        for b in p.dws:
            for q in range(b['od'] // 4, (b['od'] + _cs4(b['m'])) // 4):
                gd.ks[q] = b['kd']
        o = p.dw_in0
        L.call('cat_dwm_fwd', C.byref(gd), C.c_void_p(z1.data_ptr() + 4 * o), C.c_void_p(st1[0][0].data_ptr() + 4 * o),
               C.c_void_p(st1[0][1].data_ptr() + 4 * o), ops._p(p.w25), ops._p(p.biasd) if p.has_biasd else None, ops._p(zd), ops._p(partd), ops._stream())
        std = _finalize(p, partd, p.hcd, n, h, w, p.gammad, p.betad, [(b['od'], b['m'], b['bn2']) for b in p.dws])
    # ---- stage 2: the branch sum, K-concatenated, normalise + activation applied while staging
    segs = []
    for b in p.branches:
        if b['kind'] == 'res':
            k = b['k']
            segs.append(tconv.Segment(None, k, (k - 1) // 2, p.reflect and k > 1, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hc1,
                                      ptr=z1.data_ptr() + 4 * b['o1'], scale=st1[0][0].data_ptr() + 4 * b['o1'], shift=st1[0][1].data_ptr() + 4 * b['o1'],
                                      act=p.act, slope=p.slope, sstride=sstride_of(p.hc1)))
        else:
            segs.append(tconv.Segment(None, 1, 0, False, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hcd, ptr=zd.data_ptr() + 4 * b['od'],
                                      scale=std[0][0].data_ptr() + 4 * b['od'], shift=std[0][1].data_ptr() + 4 * b['od'], act=p.act, slope=p.slope,
                                      sstride=sstride_of(p.hcd)))
    t = ops.empty_act(n, c, h, w, dev)
    partp = torch.empty((tiles, 2, p.cs), device=dev, dtype=torch.float32)
    tconv.run(segs, p.pack2, p.bias2 if p.has_bias2 else None, t, c, n, h, w, h, w, stats=partp, scs=p.cs)
    pw = block.pw_bn
    stp = _finalize(p, partp, p.cs, n, h, w, pw.weight, pw.bias, [(0, c, pw)], mstride=c)
    # ---- out = x + pw_bn(T)
    y = ops.empty_act(n, c, h, w, dev)
    G = n if p.instance else 1
    L.call('cat_affine_res_fwd', ops._p(t), p.cs, ops._p(stp[0][0]), ops._p(stp[0][1]), sstride_of(p.cs), ops._p(x), ops.act_cs(x), ops._p(y), p.cs, G,
           (n // G) * h * w, p.cs, L.ACT_NONE, 0.0, ops._stream())
    if save is not None:
        save.update(plan=p, z1=z1, zd=zd, t=t, st1=st1, std=std, stp=stp)
    return y



# ---------------------------------------------------------------------------------------------------------------- backward
def _norm_bwd(p, n, hw, c, cs, x, dy, gamma, beta, mr, act, slope, dgamma, dbeta, accumulate=0):
    """cat_norm_bwd over (a concatenation of) train-mode norms: dx, and d gamma / d beta into the given buffers."""
    g = L.NormGeom(n, hw, c, cs, L.NORM_INSTANCE if p.instance else L.NORM_BATCH, p.eps, p.momentum, act, slope)
    dx = torch.empty((n, hw, cs), device=x.device, dtype=torch.float32)
    ws = ops.workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
    L.call('cat_norm_bwd', C.byref(g), ops._p(x), ops._p(dy), ops._p(gamma), ops._p(beta), ops._p(mr[0]), ops._p(mr[1]), ops._p(dx), ops._p(dgamma),
           ops._p(dbeta), accumulate, ops._p(ws), ops._stream())
    return dx


def _channel_sum(src, m_pix, c, cs, dst):
    ws = ops.workspace(L.query('cat_channel_sum_ws_bytes', m_pix, cs), src.device)
    L.call('cat_channel_sum', ops._p(src), m_pix, c, cs, ops._p(dst), 0, ops._p(ws), ops._stream())


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, block, *params):
        x = ops.conform(x)
        save = {}
        y = forward(block, x, save)
        ctx.block, ctx.plan = block, save['plan']
        ctx.has_dw = save['zd'] is not None
        tensors = [x, save['z1'], save['t'], save['st1'][0], save['st1'][1], save['stp'][0], save['stp'][1]]
        if ctx.has_dw:
            tensors += [save['zd'], save['std'][0], save['std'][1]]
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, block = ctx.plan, ctx.block
        saved = ctx.saved_tensors
        x, z1, t, ss1, mr1, ssp, mrp = saved[:7]
        zd, ssd, mrd = saved[7:] if ctx.has_dw else (None, None, None)
        dy = ops.conform(dy)
        p.prepare(backward=True)
        n, c, h, w = x.shape
        dev, hw, m_pix = x.device, h * w, n * h * w
        G = n if p.instance else 1
        sstr = lambda scs: scs if p.instance else 0
        st = ops._stream()
        grads = {}
        pad_mode = L.PAD_REFLECT if p.reflect else L.PAD_ZERO

        side = ops.SideJobs(dev)      # the temporaries its launches read (a1, ad, dt, dz1) stay referenced by this frame until side.join()

        def put(param, kernel):
            grads[id(param)] = ops._write_param_grad(param, kernel)

        def put_side(param, kernel):       # weight-gradient launches: independent of the data-gradient chain -> side streams
            def job():
                grads[id(param)] = ops._write_param_grad(param, lambda dst_, acc: kernel(dst_, acc, ops._stream()))
            side.run(job)

        # one stream (the inception distillers' default): the weight-gradient producers run back to back at the end and their partial sums
        # are reduced by ONE launch (ops.WgradBatch); with side streams on they stay separate jobs
        batch = None if ops.branch_streams_enabled() else ops.WgradBatch(dev, grads)

        def put_wgrad(param, make):      # make(dst) -> (geometry, x pointer, dy pointer)
            if batch is not None:
                return batch.add(param, make)

            def kernel(dst_, acc, sst):
                gw, xp, dyp = make(dst_)
                ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
                L.call('cat_conv2d_wgrad', C.byref(gw), xp, dyp, ops._p(dst_), acc, ops._p(ws), sst)
            put_side(param, kernel)

        def put_wgrad_into(dst, geom, xp, dyp):      # merged launches: destination = a gradient view of the plan, overwritten
            if batch is not None:
                return batch.add_into(dst, 0, geom, xp, dyp)

            def job():
                ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(geom)), dev)
                L.call('cat_conv2d_wgrad', C.byref(geom), xp, dyp, ops._p(dst), 0, ops._p(ws), ops._stream())
            side.run(job)

        # ---- 1. pw_bn: dT from dy (the skip connection's share of dy is added at the very end)
        pw = block.pw_bn
        if ops.act_cs(dy) != p.cs:
            raise RuntimeError('fused block backward: gradient pixel stride differs from the activation')
        if pw.weight is not None:
            tg, tb = ops._grad_target(pw.weight), ops._grad_target(pw.bias)
            if tg is not None and tb is not None:       # FusedAdam-owned: written (or accumulated) in place
                sg, sb = pw.weight._cat_grad_state, pw.bias._cat_grad_state
                dt = _norm_bwd(p, n, hw, c, p.cs, t, dy, pw.weight, pw.bias, mrp, L.ACT_NONE, 0.0, tg, tb, 0 if sg['fresh'] else 1)
                sg['fresh'] = sb['fresh'] = False
            else:
                dgp, dbp = torch.empty_like(pw.weight), torch.empty_like(pw.bias)
                dt = _norm_bwd(p, n, hw, c, p.cs, t, dy, pw.weight, pw.bias, mrp, L.ACT_NONE, 0.0, dgp, dbp)
                grads[id(pw.weight)], grads[id(pw.bias)] = dgp, dbp
        else:
            dt = _norm_bwd(p, n, hw, c, p.cs, t, dy, None, None, mrp, L.ACT_NONE, 0.0, None, None)
        # ---- 2. re-materialise the hidden activations (inputs of the second convs / of the depthwise convs)
        a1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
        L.call('cat_affine_res_fwd', ops._p(z1), p.hc1, ops._p(ss1[0]), ops._p(ss1[1]), sstr(p.hc1), None, 0, ops._p(a1), p.hc1, G, (n // G) * hw, p.hc1,
               p.act, p.slope, st)
        da1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
        ad = dad = None
        if ctx.has_dw:
            ad = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
            L.call('cat_affine_res_fwd', ops._p(zd), p.hcd, ops._p(ssd[0]), ops._p(ssd[1]), sstr(p.hcd), None, 0, ops._p(ad), p.hcd, G, (n // G) * hw,
                   p.hcd, p.act, p.slope, st)
            dad = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
        # ---- 3. second convs: weight gradients from (hidden activation slice, dT); input gradients into slices of dA1 / dAd
        side.fork()
        for b in p.branches:
            res = b['kind'] == 'res'
            k2, m, w1 = b['k2'], b['m'], b['w1']
            pad2 = (k2 - 1) // 2
            src, scs_, o = (a1, p.hc1, b['o1']) if res else (ad, p.hcd, b['od'])
            dst, dcs = (da1, p.hc1) if res else (dad, p.hcd)
            xptr = C.c_void_p(src.data_ptr() + 4 * o)
            mode2 = pad_mode if pad2 > 0 else L.PAD_ZERO
            conv2 = b['conv2']

            def kw(dst_, xptr=xptr, m=m, scs_=scs_, k2=k2, pad2=pad2, mode2=mode2):
                return ops._conv_geom(n, h, w, m, scs_, h, w, c, p.cs, k2, k2, 1, pad2, mode2, wcs=ops._grad_wcs(dst_)), xptr, ops._p(dt)
            if res or not p.merge2:
                put_wgrad(conv2.weight, kw)
            if not res and p.dpack2_dw is not None:
                continue            # input gradient: the merged launch below
            seg_pad = k2 - 1 - (0 if mode2 == L.PAD_REFLECT else pad2)
            seg = tconv.Segment(None, k2, seg_pad, False, b['d2off'], c4=p.cs, cin=c, xcs=p.cs, ptr=dt.data_ptr())
            if mode2 == L.PAD_REFLECT:
                dxp = torch.empty((n, h + 2 * pad2, w + 2 * pad2, w1), device=dev, dtype=torch.float32)
                tconv.run([seg], p.dpack2, None, dxp, m, n, h, w, h + 2 * pad2, w + 2 * pad2, ycs=w1, ycw=w1, yptr=dxp.data_ptr())
                L.call('cat_reflect_pad_bwd2', ops._p(dxp), w1, C.c_void_p(dst.data_ptr() + 4 * o), dcs, None, 0, n, h, w, w1, pad2, st)
            else:
                tconv.run([seg], p.dpack2, None, None, m, n, h, w, h, w, ycs=dcs, ycw=w1, yptr=dst.data_ptr() + 4 * o)
        if ctx.has_dw and p.dpack2_dw is not None:
            seg = tconv.Segment(None, 1, 0, False, 0, c4=p.cs, cin=c, xcs=p.cs, ptr=dt.data_ptr())
            tconv.run([seg], p.dpack2_dw, None, None, p.hcd, n, h, w, h, w, ycs=p.hcd, ycw=p.hcd, yptr=dad.data_ptr(), nvalid=sum(b['m'] for b in p.dws))
        if p.merge2:      # d W2 of all depthwise branches: dT^T x Ad as ONE 1x1 weight-gradient launch over the concatenated hidden buffer
            put_wgrad_into(p.gv['w2'], ops._conv_geom(n, h, w, p.hcd, p.hcd, h, w, c, p.cs, 1, 1, 1, 0, L.PAD_ZERO, wcs=p.hcd), ops._p(ad), ops._p(dt))
        if p.has_bias2:
            _channel_sum(dt, m_pix, c, p.cs, p.gv['c2'])
        # ---- 4. / 5. depthwise stage
        if ctx.has_dw:
            dzd = _norm_bwd(p, n, hw, p.hcd, p.hcd, zd, dad, p.gammad if p.affine else None, p.betad if p.affine else None, mrd, p.act, p.slope,
                            p.gv['gd'] if p.affine else None, p.gv['bd'] if p.affine else None)
            if p.has_biasd:
                _channel_sum(dzd, m_pix, p.hcd, p.hcd, p.gv['cd'])
            # all depthwise convs at once: input gradient (reflect padding folded in the kernel) into the dw slices of dA1, filter gradients
            # reduced straight into the parameters' gradient buffers
            nb = len(p.dws)
            gd = L.DwmGeom()
            gd.N, gd.H, gd.W, gd.nq, gd.xcs, gd.ycs, gd.scs = n, h, w, p.hcd // 4, p.hc1, p.hcd, p.hcd
            gd.reflect = int(p.reflect)
            for b in p.dws:
                for q in range(b['od'] // 4, (b['od'] + _cs4(b['m'])) // 4):
                    gd.ks[q] = b['kd']
            wts = [b['dconv'].weight for b in p.dws]
            tg = [ops._grad_target(q) for q in wts]
            if all(t_ is not None for t_ in tg):
                fresh = {q._cat_grad_state['fresh'] for q in wts}
                if len(fresh) != 1:
                    raise RuntimeError('fused block backward: depthwise gradient buffers out of sync')
                acc_dw, dsts = (0 if fresh.pop() else 1), tg
                for q in wts:
                    q._cat_grad_state['fresh'] = False
                    grads[id(q)] = None
            else:
                acc_dw, dsts = 0, [torch.empty_like(q) for q in wts]
                for q, d_ in zip(wts, dsts):
                    tq = ops._grad_target(q)
                    if tq is None:
                        grads[id(q)] = d_
            IA = C.c_int * nb
            wsd = ops.workspace(L.query('cat_dwm_bwd_ws_bytes', C.byref(gd)), dev)
            L.call('cat_dwm_bwd', C.byref(gd), C.c_void_p(a1.data_ptr() + 4 * p.dw_in0), ops._p(dzd), ops._p(p.w25),
                   C.c_void_p(da1.data_ptr() + 4 * p.dw_in0), p.hc1, nb, IA(*[b['od'] for b in p.dws]), IA(*[b['m'] for b in p.dws]),
                   IA(*[b['kd'] for b in p.dws]), (C.c_void_p * nb)(*[d_.data_ptr() for d_ in dsts]), acc_dw, ops._p(wsd), st)
            if not all(t_ is not None for t_ in tg):          # mixed ownership (tests): deliver into the owned views by hand
                for q, d_ in zip(wts, dsts):
                    tq = ops._grad_target(q)
                    if tq is not None:
                        (tq.copy_ if q._cat_grad_state['fresh'] else tq.add_)(d_)
                        q._cat_grad_state['fresh'] = False
                        grads[id(q)] = None
        # ---- 6. stage-1 norms (all branches at once)
        dz1 = _norm_bwd(p, n, hw, p.hc1, p.hc1, z1, da1, p.gamma1 if p.affine else None, p.beta1 if p.affine else None, mr1, p.act, p.slope,
                        p.gv['g1'] if p.affine else None, p.gv['b1'] if p.affine else None)
        if p.has_bias1:
            _channel_sum(dz1, m_pix, p.hc1, p.hc1, p.gv['c1'])
        # ---- 7. first convs: weight gradients from (x, dZ1 slice)
        side.refork()
        if p.merge1 is not None:
            g1 = p.merge1

            put_wgrad_into(p.gv['w1'], ops._conv_geom(n, h, w, c, p.cs, h, w, g1['width'], p.hc1, 1, 1, 1, 0, L.PAD_ZERO, wcs=p.cs), ops._p(x),
                           C.c_void_p(dz1.data_ptr() + 4 * g1['off']))
        for b in p.branches:
            if p.merge1 is not None and b['k'] == 1:
                continue
            k, m = b['k'], b['m']
            pad1 = (k - 1) // 2
            mode1 = pad_mode if pad1 > 0 else L.PAD_ZERO
            dyp = C.c_void_p(dz1.data_ptr() + 4 * b['o1'])

            def kw1(dst_, dyp=dyp, m=m, k=k, pad1=pad1, mode1=mode1):
                return ops._conv_geom(n, h, w, c, p.cs, h, w, m, p.hc1, k, k, 1, pad1, mode1, wcs=ops._grad_wcs(dst_)), ops._p(x), dyp
            put_wgrad(b['conv1'].weight, kw1)
        if batch is not None:
            batch.flush()
        # ---- 8. first convs: the six input gradients as ONE K-concatenated launch (+ the skip connection's gradient)
        dx = None
        if ctx.needs_input_grad[0]:
            M = max((b['k'] - 1) // 2 for b in p.branches) if p.reflect else 0
            segs = []
            for b in p.branches:
                pad1 = (b['k'] - 1) // 2
                segs.append(tconv.Segment(None, b['k'], (M + pad1) if M else pad1, False, b['d1off'], c4=b['w1'], cin=b['m'], xcs=p.hc1,
                                          ptr=dz1.data_ptr() + 4 * b['o1']))
            dx = ops.empty_act(n, c, h, w, dev)
            if M:
                dxp = torch.empty((n, h + 2 * M, w + 2 * M, p.cs), device=dev, dtype=torch.float32)
                tconv.run(segs, p.dpack1, None, dxp, c, n, h, w, h + 2 * M, w + 2 * M, ycs=p.cs, ycw=p.cs, yptr=dxp.data_ptr())
                L.call('cat_reflect_pad_bwd2', ops._p(dxp), p.cs, ops._p(dx), p.cs, ops._p(dy), ops.act_cs(dy), n, h, w, p.cs, M, st)
            else:
                tconv.run(segs, p.dpack1, None, dx, c, n, h, w, h, w, res=dy)
        side.join()
        # ---- 9. scatter the concatenated parameter gradients
        all_t = [q for _, _, _, q in p.targets] + [t2[5] for t2 in p.targets2d]
        owned = [getattr(q, '_cat_grad_view', None) is not None for q in all_t]
        if all_t and all(owned):
            fresh = {q._cat_grad_state['fresh'] for q in all_t}
            if len(fresh) != 1:
                raise RuntimeError('fused block backward: gradient buffers of one block out of sync')
            views = tuple(q._cat_grad_view.data_ptr() for q in all_t)
            if p.scatter_jobs is None or p.scatter_jobs[3] != views:
                jobs = [dict(kind=3, srcs=[p.gv[v].data_ptr() + 4 * o, q._cat_grad_view.data_ptr()], nsrc=2, n=cnt, threads=cnt) for v, o, cnt, q in p.targets]
                # a one-channel conv weight is stored unpadded (wcs 1): never more columns than the destination row holds
                for v, o, rows, cols, sstr_, q in p.targets2d:
                    wcs_q = ops._grad_wcs(q._cat_grad_view)
                    cq = min(cols, wcs_q)
                    jobs.append(dict(kind=4, srcs=[p.gv[v].data_ptr() + 4 * o, q._cat_grad_view.data_ptr()], nsrc=2, n=rows * cq, cs=cq, wn=sstr_,
                                     wcs=wcs_q, threads=rows * cq))
                p.scatter_jobs = p._jobs_to_dev(jobs) + (views,)
            tj, nj, nb, _ = p.scatter_jobs
            L.call('cat_prep_run', ops._p(tj), nj, nb, 0 if fresh.pop() else 1, st)
            for q in all_t:
                q._cat_grad_state['fresh'] = False
                grads[id(q)] = None
        else:
            def deliver(q, gq):
                tgt = getattr(q, '_cat_grad_view', None)
                if tgt is not None:
                    stq = q._cat_grad_state
                    (tgt.copy_ if stq['fresh'] else tgt.add_)(gq)
                    stq['fresh'] = False
                    gq = None
                grads[id(q)] = gq
            for v, o, cnt, q in p.targets:
                flat = p.gv[v][o:o + cnt]
                if q.dim() == 4:      # rows of a merged weight gradient: back into the parameter's [O][kh][kw][wcs] storage
                    gq = ops.padded_weight_like(q.shape, dev)
                    torch.as_strided(gq, (cnt,), (1,), gq.storage_offset()).copy_(flat)
                else:
                    gq = flat.clone()
                deliver(q, gq)
            for v, o, rows, cols, sstr_, q in p.targets2d:
                gq = ops.padded_weight_like(q.shape, dev)
                cols = min(cols, ops.weight_wcs(gq))
                src2 = torch.as_strided(p.gv[v], (rows, cols), (sstr_, 1), o)
                torch.as_strided(gq, (rows, cols), (ops.weight_wcs(gq), 1), gq.storage_offset()).copy_(src2)
                deliver(q, gq)
        return (dx, None) + tuple(grads.get(id(q)) for q in block.parameters())


def apply(block, x):
    if not torch.is_grad_enabled():
        return forward(block, ops.conform(x))
    return _BlockFn.apply(x, block, *block.parameters())
