"""Fused training-mode path of the six-branch units of the GauGAN generator (reference models/modules/inception_modules.py:428-505,
549-562: SPADEInvertedResidualChannels; :648-700, 746-762: the gamma|beta net of InceptionSPADE):

    unit(x) = sum_k conv2_k(relu(bn(conv1_k(x)))) + sum_k pw2_k(relu(bn(dw_k(relu(bn(pw1_k(x))))))) [+ r]

with train-mode (Synchronized)BatchNorm2d in every position, zero ("same") padding, C_in != C_out in general and an optional addend r (the
block's shortcut).  The per-layer path launches ~45 kernels per unit (12 convs, 9 norms x 3, add_n); most of them HBM-bound passes over
hidden tensors of 1..13 channels on planes of up to 256 x 512 pixels.  Here a unit is the protocol of cat_amd/fused_block.py without the
closing pw_bn:

    stage 1   first convs of all branches -> one concatenated pre-norm buffer Z1 + per-tile statistics   (cat_tstage1_fwd / cat_tconv_fwd)
    finalize  scale / shift of all stage-1 norms + their running statistics                               (cat_tnorm_finalize)
    dw        all depthwise convs as one launch, norm + ReLU of stage 1 applied while staging             (cat_dwm_fwd)
    finalize
    stage 2   the branch sum: six second convs K-concatenated, norm + ReLU applied while staging, bias and the addend r in the epilogue

5 launches, no normalised tensor is ever written.  The backward pass re-materialises the two hidden activations and reuses the kernels of
the fused inception block (branch-wise weight gradients on side streams, norm backward once per stage over the concatenation, the first-conv
input gradients as one K-concatenated launch).  Taken when the unit runs in training mode on one rank (with several ranks the
SynchronizedBatchNorm statistics are exchanged per layer: general path), on planes of at least `ops._TCONV_MIN_TILES` 8 x 16 tiles (the
64 x 128 .. 256 x 512 stages at batch 4); tests/test_spade_gpu.py::test_fused_spade_units_match_general_path pins it to the general path."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import nn as cnn
from . import ops
from . import optim
from . import tconv

_ENABLED = os.environ.get('CAT_FUSED_SPADE', '1') != '0'      # A/B switch; 'train' / 'frozen' select one of the two forms
_ONLY = os.environ.get('CAT_FUSED_SPADE', '1') if os.environ.get('CAT_FUSED_SPADE', '1') in ('train', 'frozen') else None


_SYNC_FUSED = os.environ.get('CAT_FUSED_SPADE_SYNC', '1') != '0'      # fused units under SynchronizedBatchNorm over several ranks (round 4)
_S1_DGRAD = os.environ.get('CAT_FUSED_SPADE_S1_DGRAD', '1') != '0'     # A/B switch: the second convs' input gradients as one launch
_UNITS = os.environ.get('CAT_FUSED_SPADE_UNITS', 'all')      # A/B switch: 'gb' / 'main' = only the gamma|beta nets / only the main units
STATS = {'train_fwd': 0, 'frozen_fwd': 0, 'bwd': 0, 'collectives': 0}      # calls per form; statistics exchanges issued (tests / diagnostics)


def set_enabled(on):
    global _ENABLED
    _ENABLED = bool(on)


def _cs4(c):
    return (c + 3) // 4 * 4


def _conv_of(m):
    """nn.Conv2d behind a `Conv` wrapper (main branches) or the plain conv (gamma|beta nets)."""
    return m.conv if hasattr(m, 'conv') and not isinstance(m, cnn.Conv2d) else m


def _branches(res_ops, dw_ops):
    res = [dict(kind='res', k=op[0].conv.kernel_size[0], m=op[0].conv.out_channels, conv1=op[0].conv, bn1=op[0].norm, act=op[0].active,
                conv2=_conv_of(op[1])) for op in res_ops]
    dws = [dict(kind='dw', k=1, kd=op[1].conv.kernel_size[0], m=op[0].conv.out_channels, conv1=op[0].conv, bn1=op[0].norm, act=op[0].active,
                dconv=op[1].conv, bn2=op[1].norm, conv2=_conv_of(op[2])) for op in dw_ops]
    return res, dws


def _has_hooks(mods):
    for m in mods:
        for s in m.modules():
            if s._forward_hooks or s._forward_pre_hooks or s._backward_hooks:
                return True
    return False


def applicable(res_ops, dw_ops, x, training):
    """training: the train-mode path (batch statistics, autograd).  not training: the frozen path -- eval-mode norms under no_grad (the
    teacher): the same three kernels with scale / shift folded from the running statistics, no statistics / finalize launches."""
    if not _ENABLED or not x.is_cuda or not ops.is_act(x):
        return False
    if _ONLY is not None and _ONLY != ('train' if training else 'frozen'):
        return False
    if training and ops.bn_sync() is not None and not _SYNC_FUSED:
        return False      # A/B switch: the per-layer path with one statistics exchange per norm layer
    if not training and torch.is_grad_enabled():
        return False
    if len(res_ops) + len(dw_ops) == 0 or len(res_ops) + len(dw_ops) > L.TCONV_MAXSEG:
        return False
    n, c, h, w = x.shape
    # planes too small to fill the chip run faster layer by layer on ONE GPU (split-K im2col kernels: 62.8 vs 60.9 images/s, round 3) -- but
    # layer by layer every norm is its own statistics exchange over the ranks (18 per unit against 4): under SynchronizedBatchNorm with
    # N > 1 ranks every unit the kernels can run is fused
    synced = training and ops.bn_sync() is not None and _SYNC_FUSED
    if not synced and not ops.tconv_applicable(n, h, w, 16, 3, 3, 1, 1):
        return False
    if c == 1:      # FusedAdam stores [m][1][1][1] weights unpadded (row stride 1): the merged first-conv gradient scatter writes cs4(Cin)-wide rows
        return False
    res, dws = _branches(res_ops, dw_ops)
    first = (res + dws)[0]
    act0 = cnn._act_code(first['act'])
    for b in res + dws:
        dw = b['kind'] == 'dw'
        convs = [b['conv1'], b['conv2']] + ([b['dconv']] if dw else [])
        if any('weight_orig' in cv._parameters or cv.stride[0] != 1 or cv.dilation != (1, 1) or cv.padding_mode != 'zeros' for cv in convs):
            return False      # spectral norm / strides / dilation: general path
        if b['conv1'].groups != 1 or b['conv2'].groups != 1 or (dw and b['dconv'].groups != b['m']):
            return False
        norms = [b['bn1']] + ([b['bn2']] if dw else [])
        if any(not isinstance(nm, cnn.BatchNorm2d) or nm.training != training or not nm.track_running_stats or nm.momentum is None for nm in norms):
            return False
        # under a multi-rank reducer the fused stage finalize all-reduces the statistics and uses the clamp(var, eps) formula: that is
        # SynchronizedBatchNorm2d's arithmetic (sync_batchnorm/batchnorm.py:103-140).  A plain nn.BatchNorm2d (norm_G = 'spadebatch3x3') stays
        # per-replica in the reference (local statistics, var + eps): such units take the per-layer path, whose BatchNorm2d is never synced
        if synced and any(not isinstance(nm, cnn.SynchronizedBatchNorm2d) for nm in norms):
            return False
        # the plan keeps ONE (eps, momentum, activation) for the whole unit and assumes 'same' zero padding everywhere
        if any(float(nm.eps) != float(first['bn1'].eps) or float(nm.momentum) != float(first['bn1'].momentum) for nm in norms):
            return False
        if cnn._act_code(b['act']) != act0 or act0[0] not in (L.ACT_RELU, L.ACT_LRELU):
            return False
        k2 = 1 if dw else b['k']
        ks = [b['k'], b.get('kd', 1), b['conv2'].kernel_size[0]]
        if any(k not in (1, 3, 5) for k in ks) or b['conv2'].kernel_size[0] != k2:
            return False
        if b['conv1'].padding[0] != (b['k'] - 1) // 2 or b['conv2'].padding[0] != (k2 - 1) // 2:
            return False
        if dw and b['dconv'].padding[0] != (b.get('kd', 1) - 1) // 2:
            return False
    if sum(_cs4(b['m']) for b in dws) > 4 * (L.DWM_MAXQ_BWD if training else L.DWM_MAXQ):
        return False
    return not _has_hooks(list(res_ops) + list(dw_ops))


class _Plan:
    """Static layout of one unit: channel slices, persistent operand buffers and the preparation job table (cf. fused_block._Plan)."""

    def __init__(self, res_ops, dw_ops, cin, cout, dev):
        self.dev = dev
        res, dws = _branches(res_ops, dw_ops)
        self.mods = list(res_ops) + list(dw_ops)
        for b in res + dws:
            cnn._to_channels_last_(b['conv1'])
            cnn._to_channels_last_(b['conv2'])
        self.Cin, self.csi = cin, _cs4(cin)
        self.Cout, self.cso = cout, _cs4(cout)
        first = (res + dws)[0]
        self.act, self.slope = cnn._act_code(first['act'])
        self.eps, self.momentum = float(first['bn1'].eps), float(first['bn1'].momentum)
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
        self.groups = []
        for k in (1, 3, 5):
            bs = [b for b in order if b['k'] == k]
            if bs:
                g0, g1 = bs[0]['o1'], bs[-1]['o1'] + bs[-1]['w1']
                self.groups.append(dict(k=k, off=g0, width=g1 - g0, branches=bs))
        z = lambda n: torch.zeros(max(n, 4), device=dev, dtype=torch.float32)
        for g in self.groups:
            g['pack'] = z(tconv.pack_floats(g['k'], self.csi, g['width']))
        # non-affine norms (the depthwise branches' second norm of a SPADE block) read gamma = 1 / beta = 0 from the concatenated vectors
        self.gamma1, self.beta1, self.bias1 = torch.ones(max(self.hc1, 4), device=dev), z(self.hc1), z(self.hc1)
        self.gammad, self.betad, self.biasd = torch.ones(max(self.hcd, 4), device=dev), z(self.hcd), z(self.hcd)
        self.bias2 = z(self.cso)
        self.w25 = z(25 * max(self.hcd, 4))
        self.has_bias1 = any(b['conv1'].bias is not None for b in order)
        self.has_biasd = any(b['dconv'].bias is not None for b in dws)
        self.has_bias2 = any(b['conv2'].bias is not None for b in order)
        po = 0
        for b in order:
            k2 = b['k'] if b['kind'] == 'res' else 1
            b['k2'], b['p2off'] = k2, po
            po += tconv.pack_floats(k2, b['w1'], self.Cout)
        self.pack2 = z(po)
        po = 0
        for b in order:
            b['d2off'] = po
            po += tconv.pack_floats(b['k2'], self.cso, b['m'])
        self.dpack2 = z(po)
        # the second convs' input gradients as ONE launch (cat_tstage1_dgrad: dT is staged once for the 5 x 5 and the 3 x 3 residual branch
        # and the N-concatenated 1 x 1 second convs of the depthwise branches) where a kernel exists for the widths
        r5, r3 = [b for b in res if b['k'] == 5], [b for b in res if b['k'] == 3]
        self.s1d = None
        if _S1_DGRAD and len(r5) == 1 and len(r3) == 1 and dws and L.query('cat_tstage1_dgrad_supported', r5[0]['w1'], r3[0]['w1'], self.hcd):
            self.s1d = (r5[0], r3[0])
            self.dpack2_dw = z(tconv.pack_floats(1, self.cso, self.hcd))
        po = 0
        for b in order:
            b['d1off'] = po
            po += tconv.pack_floats(b['k'], b['w1'], self.Cin)
        self.dpack1 = z(po)
        self.gv = dict(g1=z(self.hc1), b1=z(self.hc1), c1=z(self.hc1), gd=z(self.hcd), bd=z(self.hcd), cd=z(self.hcd), c2=z(self.cso))
        self.targets = []       # (vector name, offset, n, parameter)
        for b in order:
            if b['bn1'].weight is not None:
                self.targets += [('g1', b['o1'], b['m'], b['bn1'].weight), ('b1', b['o1'], b['m'], b['bn1'].bias)]
            if b['conv1'].bias is not None:
                self.targets.append(('c1', b['o1'], b['m'], b['conv1'].bias))
            if b['conv2'].bias is not None:
                self.targets.append(('c2', 0, self.Cout, b['conv2'].bias))
        for b in dws:
            if b['bn2'].weight is not None:
                self.targets += [('gd', b['od'], b['m'], b['bn2'].weight), ('bd', b['od'], b['m'], b['bn2'].bias)]
            if b['dconv'].bias is not None:
                self.targets.append(('cd', b['od'], b['m'], b['dconv'].bias))
        # merged weight-gradient launches (as in fused_block): the 1 x 1 first convs of all branches are ONE GEMM over the N-concatenated dZ1
        # slice (x is read once instead of once per branch; rows of `w1` then go to the parameters), the 1 x 1 second convs of the depthwise
        # branches one K-concatenated GEMM over the whole depthwise hidden buffer (columns of `w2`)
        g1 = next((g for g in self.groups if g['k'] == 1), None)
        self.merge1 = g1 if (g1 is not None and len(g1['branches']) > 1) else None
        if self.merge1 is not None:
            self.gv['w1'] = z(g1['width'] * self.csi)
            for b in g1['branches']:
                self.targets.append(('w1', (b['o1'] - g1['off']) * self.csi, b['m'] * self.csi, b['conv1'].weight))
        self.merge2 = len(dws) > 1
        self.targets2d = []      # (vector, src offset, rows, cols, src stride, parameter): dst stride = the parameter's own wcs
        if self.merge2:
            self.gv['w2'] = z(self.Cout * self.hcd)
            for b in dws:
                self.targets2d.append(('w2', b['od'], self.Cout, _cs4(b['m']), self.hcd, b['conv2'].weight))
        self.params = []
        seen = set()
        for m in self.mods:
            for q in m.parameters():
                if id(q) not in seen:
                    seen.add(id(q))
                    self.params.append(q)
        self.scatter_jobs = None
        self._build_jobs()
        self.key = self.bkey = None

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
        t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        self._last_arr = arr      # host copy (fused_block.prepare_plans merges the tables of all units of a generator into one launch)
        return t, len(jobs), blk

    def _pack_job(self, w, dst_ptr, mode, nn, ck, ks, nt_total, col0):
        wcl, wcs = ops.weight_cl(w)
        if wcl.data_ptr() != w.data_ptr():
            raise RuntimeError('fused SPADE unit: conv weights must be in kernel layout')
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
                fwd.append(self._pack_job(b['conv1'].weight, g['pack'].data_ptr(), tconv.FWD, b['m'], self.Cin, g['k'], nt, b['o1'] - g['off']))
        nt2 = (self.Cout + 15) // 16
        nt1 = (self.Cin + 15) // 16
        for b in self.branches:
            if b['bn1'].weight is not None:
                fwd.append(vec(self.gamma1, b['o1'], [b['bn1'].weight], b['m']))
                fwd.append(vec(self.beta1, b['o1'], [b['bn1'].bias], b['m']))
            if b['conv1'].bias is not None:
                fwd.append(vec(self.bias1, b['o1'], [b['conv1'].bias], b['m']))
            fwd.append(self._pack_job(b['conv2'].weight, self.pack2.data_ptr() + 4 * b['p2off'], tconv.FWD, self.Cout, b['m'], b['k2'], nt2, 0))
            bwd.append(self._pack_job(b['conv2'].weight, self.dpack2.data_ptr() + 4 * b['d2off'], tconv.DGRAD, b['m'], self.Cout, b['k2'], (b['m'] + 15) // 16, 0))
            bwd.append(self._pack_job(b['conv1'].weight, self.dpack1.data_ptr() + 4 * b['d1off'], tconv.DGRAD, self.Cin, b['m'], b['k'], nt1, 0))
            if self.s1d is not None and b['kind'] == 'dw':
                bwd.append(self._pack_job(b['conv2'].weight, self.dpack2_dw.data_ptr(), tconv.DGRAD, b['m'], self.Cout, 1, (self.hcd + 15) // 16, b['od']))
        for b in self.dws:
            if b['bn2'].weight is not None:
                fwd.append(vec(self.gammad, b['od'], [b['bn2'].weight], b['m']))
                fwd.append(vec(self.betad, b['od'], [b['bn2'].bias], b['m']))
            if b['dconv'].bias is not None:
                fwd.append(vec(self.biasd, b['od'], [b['dconv'].bias], b['m']))
            kd = b['kd']
            wd = b['dconv'].weight
            if not wd.is_contiguous():
                raise RuntimeError('fused SPADE unit: depthwise weights must be contiguous')
            fwd.append(dict(kind=2, srcs=[wd.data_ptr()], dst=self.w25.data_ptr(), Nn=b['m'], ks=kd, col0=b['od'], cs=self.hcd, threads=b['m'] * kd * kd))
        b2 = [b['conv2'].bias for b in self.branches if b['conv2'].bias is not None]
        if b2:
            fwd.append(vec(self.bias2, 0, b2, self.Cout))
        self.ptrs = tuple(q.data_ptr() for q in self.params)
        self.shapes = tuple(tuple(q.shape) for q in self.params)
        self.ids = tuple(id(q) for q in self.params)
        self.fwd_jobs = self._jobs_to_dev(fwd)
        self.fwd_arr = self._last_arr
        self.bwd_jobs = self._jobs_to_dev(bwd)
        self.bwd_arr = self._last_arr
        self.tables_version = getattr(self, 'tables_version', 0) + 1

    def __deepcopy__(self, memo):
        """A copied module builds its own plan at its first forward (cf. fused_block._Plan.__deepcopy__)."""
        return None

    def _epoch_key(self):
        trainable = any(getattr(q, '_cat_grad_view', None) is not None for q in self.params)
        return (optim.weights_epoch() if trainable else -1, tuple(q._version for q in self.params))

    def ptrs_now(self):
        return tuple(q.data_ptr() for q in self.params)

    def prepare(self, backward=False):
        gref = getattr(self, 'group', None)      # weak reference to the generator that owns this unit (no cycle through the cached plans)
        group = gref() if gref is not None else None
        if group is not None and (self.bkey if backward else self.key) != self._epoch_key():
            # the first stale unit of a pass refreshes the operands of EVERY fused unit of the generator in one launch (round 4)
            from . import fused_block
            fused_block.prepare_plans(units_of(group), gref, backward)
        if tuple(q.data_ptr() for q in self.params) != self.ptrs:       # FusedAdam re-housed the parameters: same layout, new addresses
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


def units_of(generator):
    """Plans of every fused unit (main six-branch units and gamma|beta nets) built so far under `generator`, in module order."""
    out = []
    for m in generator.modules():
        for slot in ('_cat_fused_main', '_cat_fused_gb'):
            p = m.__dict__.get(slot)
            if p is not None:
                out.append(p)
    return out


PLAN_GEN = 0      # bumped whenever a unit plan is (re)built: generators regroup their units' operand preparation when it moves


def plan_for(owner, slot, res_ops, dw_ops, cin, cout, x):
    global PLAN_GEN
    p = getattr(owner, slot, None)
    params = [q for m in list(res_ops) + list(dw_ops) for q in m.parameters()]
    if p is None or p.dev != x.device or p.shapes != tuple(tuple(q.shape) for q in params) or p.ids != tuple(id(q) for q in params):
        p = _Plan(res_ops, dw_ops, cin, cout, x.device)
        setattr(owner, slot, p)
        PLAN_GEN += 1
    return p


def _slices(pairs):
    arr = (L.NSlice * len(pairs))()
    for i, (c0, c, bn) in enumerate(pairs):
        arr[i].c0, arr[i].c = c0, c
        arr[i].running_mean = bn.running_mean.data_ptr()
        arr[i].running_var = bn.running_var.data_ptr()
        # SynchronizedBatchNorm2d.forward bypasses _BatchNorm.forward: num_batches_tracked is never advanced (batchnorm.py:68-101)
        arr[i].num_batches = None if isinstance(bn, cnn.SynchronizedBatchNorm2d) else bn.num_batches_tracked.data_ptr()
    return arr


def _finalize_g(p, part, scs, n, h, w, gamma, beta, pairs):
    """(scale | shift), (mean | rstd) of every norm of a stage.  With a SynchronizedBatchNorm reducer installed (N > 1 ranks) the stage's
    statistics are exchanged ONCE: this rank's tile table -> [sum x | sum x^2] of the whole concatenation -> one all-reduce -> the
    reference's multi-replica formula (batchnorm.py:103-140: clamp(var, eps), unbiased running_var); the second tensor then holds
    (a | b) = (inv_std | -mean * inv_std) for the split-phase backward.
    A GENERATOR: it yields the tensor to be sum-reduced over the ranks and continues once that has happened -- `_drive` performs the exchange
    on the spot (one unit), `_drive_many` runs several independent units in lockstep and merges their k-th exchanges into ONE collective
    (the gamma|beta nets of all SPADE layers of a generator: `prepass`)."""
    ss = torch.empty((2, 1, scs), device=part.device, dtype=torch.float32)
    mr = torch.empty((2, 1, scs), device=part.device, dtype=torch.float32)
    sync = ops.bn_sync()
    if sync is None:
        L.call('cat_tnorm_finalize', ops._p(part), scs, 1, n, h, w, ops._p(gamma), ops._p(beta), len(pairs), _slices(pairs), p.eps, p.momentum,
               ops._p(ss[0]), ops._p(ss[1]), ops._p(mr[0]), ops._p(mr[1]), scs, ops._stream())
        return ss, mr
    sums = yield Alloc(2 * scs, part.device)      # a slice of the round's arena under _drive_many: the merged exchange needs no pack / unpack
    L.call('cat_tnorm_sums', ops._p(part), scs, n, h, w, 8, 16, 1, ops._p(sums), ops._stream())
    yield sums
    count = float(n * h * w) * sync.world_size
    L.call('cat_tnorm_finalize_sums', ops._p(sums), count, scs, ops._p(gamma), ops._p(beta), len(pairs), _slices(pairs), p.eps, p.momentum, 1,
           ops._p(ss[0]), ops._p(ss[1]), ops._p(mr[0]), ops._p(mr[1]), ops._stream())
    return ss, mr


class Alloc:
    """A unit generator's request for the buffer of its next statistics exchange (n floats, n % 4 == 0).  `_drive_many` hands every unit of
    a lockstep round a slice of ONE arena, in unit order, so that the round's exchanges are one contiguous message: the collective runs on
    the arena itself, without the torch.cat / copy_ kernels a pack + unpack would launch (round-5 verdict, robustness #15)."""
    __slots__ = ('n', 'device')

    def __init__(self, n, device):
        self.n, self.device = int(n), device


def _drive(gen):
    """Run ONE unit generator to completion; every statistics exchange it asks for happens immediately (one collective each)."""
    try:
        req = next(gen)
        while True:
            if isinstance(req, Alloc):
                req = gen.send(torch.empty(req.n, device=req.device, dtype=torch.float32))
                continue
            ops.bn_sync().all_reduce_sum_(req)
            STATS['collectives'] += 1
            req = gen.send(None)
    except StopIteration as e:
        return e.value


def _drive_many(gens):
    """Run several INDEPENDENT unit generators in lockstep: the k-th exchanges of all of them travel as ONE collective (their buffers are
    summed element-wise either way: the same arithmetic per element as `_drive` on each; the reduction order is the backend's).  Units that
    ask for their buffer first (`Alloc`) get adjacent slices of one arena, and the collective is issued on the arena."""
    results, reqs = [None] * len(gens), {}
    for i, g in enumerate(gens):
        try:
            reqs[i] = next(g)
        except StopIteration as e:
            results[i] = e.value
    while reqs:
        allocs = [i for i in sorted(reqs) if isinstance(reqs[i], Alloc)]
        if allocs:
            dev = reqs[allocs[0]].device
            sizes = [(reqs[i].n + 3) // 4 * 4 for i in allocs]
            arena = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            o = 0
            for i, sz in zip(allocs, sizes):
                n = reqs[i].n
                try:
                    reqs[i] = gens[i].send(arena[o:o + n])
                except StopIteration as e:
                    results[i] = e.value
                    del reqs[i]
                o += sz
            continue
        order = sorted(reqs)
        ops.bn_sync().all_reduce_sum_many_([reqs[i] for i in order])
        STATS['collectives'] += 1
        nxt = {}
        for i in order:
            try:
                nxt[i] = gens[i].send(None)
            except StopIteration as e:
                results[i] = e.value
        reqs = nxt
    return results


def forward(p, x, addend, save=None):
    """The unit's forward.  `addend`: NHWC activation [n, Cout, h, w] added in the epilogue (the block's shortcut) or None."""
    return _drive(forward_g(p, x, addend, save))


def forward_g(p, x, addend, save=None):
    """forward as a generator over its statistics exchanges (see _finalize_g)."""
    STATS['train_fwd'] += 1
    p.prepare()
    n, c, h, w = x.shape
    dev = x.device
    tiles = n * ((h + 7) // 8) * ((w + 15) // 16)
    z1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
    part1 = torch.empty((tiles, 2, p.hc1), device=dev, dtype=torch.float32)
    by_k = {g['k']: g for g in p.groups}
    if len(p.groups) == 3 and L.query('cat_tstage1_supported', by_k[5]['width'], by_k[3]['width'], by_k[1]['width']):
        gs = L.Stage1Geom()
        gs.N, gs.H, gs.W, gs.xcs, gs.cin, gs.reflect, gs.ycs, gs.scs = n, h, w, ops.act_cs(x), c, 0, p.hc1, p.hc1
        packs = (C.c_void_p * 3)()
        for slot, k in enumerate((5, 3, 1)):
            g = by_k[k]
            gs.col0[slot], gs.width[slot], gs.nvalid[slot] = g['off'], g['width'], sum(b['m'] for b in g['branches'])
            packs[slot] = g['pack'].data_ptr()
        L.call('cat_tstage1_fwd', C.byref(gs), ops._p(x), packs, ops._p(p.bias1) if p.has_bias1 else None, ops._p(z1), ops._p(part1), ops._stream())
    else:
        for g in p.groups:
            pad = (g['k'] - 1) // 2
            seg = tconv.Segment(x, g['k'], pad, False, 0)
            tconv.run([seg], g['pack'], (p.bias1.data_ptr() + 4 * g['off']) if p.has_bias1 else None, None, g['width'], n, h, w, h, w, ycs=p.hc1,
                      ycw=g['width'], yptr=z1.data_ptr() + 4 * g['off'], stats=part1.data_ptr() + 4 * g['off'], scs=p.hc1,
                      nvalid=sum(b['m'] for b in g['branches']))
    st1 = yield from _finalize_g(p, part1, p.hc1, n, h, w, p.gamma1, p.beta1, [(b['o1'], b['m'], b['bn1']) for b in p.branches])
    zd = std = None
    if p.dws:
        zd = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
        partd = torch.empty((tiles, 2, p.hcd), device=dev, dtype=torch.float32)
        gd = L.DwmGeom()
        gd.N, gd.H, gd.W, gd.nq, gd.xcs, gd.ycs, gd.scs = n, h, w, p.hcd // 4, p.hc1, p.hcd, p.hcd
        gd.sstride, gd.reflect, gd.act, gd.slope = 0, 0, p.act, p.slope
        for b in p.dws:
            for q in range(b['od'] // 4, (b['od'] + _cs4(b['m'])) // 4):
                gd.ks[q] = b['kd']
        o = p.dw_in0
        L.call('cat_dwm_fwd', C.byref(gd), C.c_void_p(z1.data_ptr() + 4 * o), C.c_void_p(st1[0][0].data_ptr() + 4 * o),
               C.c_void_p(st1[0][1].data_ptr() + 4 * o), ops._p(p.w25), ops._p(p.biasd) if p.has_biasd else None, ops._p(zd), ops._p(partd), ops._stream())
        std = yield from _finalize_g(p, partd, p.hcd, n, h, w, p.gammad, p.betad, [(b['od'], b['m'], b['bn2']) for b in p.dws])
    segs = []
    for b in p.branches:
        if b['kind'] == 'res':
            k = b['k']
            segs.append(tconv.Segment(None, k, (k - 1) // 2, False, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hc1, ptr=z1.data_ptr() + 4 * b['o1'],
                                      scale=st1[0][0].data_ptr() + 4 * b['o1'], shift=st1[0][1].data_ptr() + 4 * b['o1'], act=p.act, slope=p.slope))
        else:
            segs.append(tconv.Segment(None, 1, 0, False, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hcd, ptr=zd.data_ptr() + 4 * b['od'],
                                      scale=std[0][0].data_ptr() + 4 * b['od'], shift=std[0][1].data_ptr() + 4 * b['od'], act=p.act, slope=p.slope))
    y = ops.empty_act(n, p.Cout, h, w, dev)
    tconv.run(segs, p.pack2, p.bias2 if p.has_bias2 else None, y, p.Cout, n, h, w, h, w, res=addend)
    if save is not None:
        save.update(z1=z1, zd=zd, st1=st1, std=std)
    return y


def _eval_affine(p):
    """scale / shift of every norm from its running statistics (eval mode), concatenated like the train-mode finalize output; cached until
    a tensor involved changes (the frozen teacher: once)."""
    tensors = []
    for b in p.branches:
        tensors += [b['bn1'].weight, b['bn1'].bias, b['bn1'].running_mean, b['bn1'].running_var]
    for b in p.dws:
        tensors += [b['bn2'].weight, b['bn2'].bias, b['bn2'].running_mean, b['bn2'].running_var]
    trainable = any(getattr(t, '_cat_grad_view', None) is not None for t in tensors if t is not None)
    key = (tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors), optim.weights_epoch() if trainable else -1)
    cached = getattr(p, 'eval_ss', None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    dev = p.dev
    ss1 = torch.zeros((2, max(p.hc1, 4)), device=dev, dtype=torch.float32)
    ssd = torch.zeros((2, max(p.hcd, 4)), device=dev, dtype=torch.float32)
    st = ops._stream()

    def fold(bn, dst, off, m):
        L.call('cat_bn_fold', ops._p(bn.weight), ops._p(bn.bias), ops._p(bn.running_mean), ops._p(bn.running_var), float(bn.eps), m,
               C.c_void_p(dst[0].data_ptr() + 4 * off), C.c_void_p(dst[1].data_ptr() + 4 * off), st)
    for b in p.branches:
        fold(b['bn1'], ss1, b['o1'], b['m'])
    for b in p.dws:
        fold(b['bn2'], ssd, b['od'], b['m'])
    p.eval_ss = (key, ss1, ssd)
    return ss1, ssd


def forward_eval(p, x, addend):
    """The unit with eval-mode norms (no grad): stage 1 -> depthwise stage -> branch sum, the norms folded into the consumers' staging."""
    STATS['frozen_fwd'] += 1
    p.prepare()
    ss1, ssd = _eval_affine(p)
    n, c, h, w = x.shape
    dev = x.device
    z1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
    for g in p.groups:
        pad = (g['k'] - 1) // 2
        seg = tconv.Segment(x, g['k'], pad, False, 0)
        tconv.run([seg], g['pack'], (p.bias1.data_ptr() + 4 * g['off']) if p.has_bias1 else None, None, g['width'], n, h, w, h, w, ycs=p.hc1,
                  ycw=g['width'], yptr=z1.data_ptr() + 4 * g['off'], nvalid=sum(b['m'] for b in g['branches']))
    zd = None
    if p.dws:
        zd = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
        gd = L.DwmGeom()
        gd.N, gd.H, gd.W, gd.nq, gd.xcs, gd.ycs, gd.scs = n, h, w, p.hcd // 4, p.hc1, p.hcd, p.hcd
        gd.sstride, gd.reflect, gd.act, gd.slope = 0, 0, p.act, p.slope
        for b in p.dws:
            for q in range(b['od'] // 4, (b['od'] + _cs4(b['m'])) // 4):
                gd.ks[q] = b['kd']
        o = p.dw_in0
        L.call('cat_dwm_fwd', C.byref(gd), C.c_void_p(z1.data_ptr() + 4 * o), C.c_void_p(ss1[0].data_ptr() + 4 * o), C.c_void_p(ss1[1].data_ptr() + 4 * o),
               ops._p(p.w25), ops._p(p.biasd) if p.has_biasd else None, ops._p(zd), None, ops._stream())
    segs = []
    for b in p.branches:
        if b['kind'] == 'res':
            k = b['k']
            segs.append(tconv.Segment(None, k, (k - 1) // 2, False, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hc1, ptr=z1.data_ptr() + 4 * b['o1'],
                                      scale=ss1[0].data_ptr() + 4 * b['o1'], shift=ss1[1].data_ptr() + 4 * b['o1'], act=p.act, slope=p.slope))
        else:
            segs.append(tconv.Segment(None, 1, 0, False, b['p2off'], c4=b['w1'], cin=b['m'], xcs=p.hcd, ptr=zd.data_ptr() + 4 * b['od'],
                                      scale=ssd[0].data_ptr() + 4 * b['od'], shift=ssd[1].data_ptr() + 4 * b['od'], act=p.act, slope=p.slope))
    y = ops.empty_act(n, p.Cout, h, w, dev)
    tconv.run(segs, p.pack2, p.bias2 if p.has_bias2 else None, y, p.Cout, n, h, w, h, w, res=addend)
    return y


def _norm_bwd_g(p, n, hw, c, cs, x, dy, gamma, beta, mr, dgamma, dbeta, synced=False):
    dx = torch.empty((n, hw, cs), device=x.device, dtype=torch.float32)
    if synced:
        # SynchronizedBatchNorm backward over ranks: local [sum g | sum g * xhat] of the whole stage -> ONE all-reduce -> apply; the parameter
        # gradients stay local sums (the gradient bucket all-reduce averages them), as in ops.SyncBNFn
        sync = ops.bn_sync()
        if sync is None:
            raise RuntimeError('fused SPADE unit backward: the forward ran under a SynchronizedBatchNorm reducer that is gone')
        m = n * hw
        sums = yield Alloc(2 * cs, x.device)
        ws = ops.workspace(L.query('cat_bn_ws_bytes', m, cs), x.device)
        st = ops._stream()
        L.call('cat_bn_stats_bwd', ops._p(x), ops._p(dy), ops._p(gamma), ops._p(beta), ops._p(mr[0]), ops._p(mr[1]), m, c, cs, p.act, p.slope,
               ops._p(sums), ops._p(ws), st)
        local = sums.clone()
        yield sums      # sum-reduced over the ranks by the driver (_drive / _drive_many)
        L.call('cat_bn_apply_bwd', ops._p(x), ops._p(dy), ops._p(gamma), ops._p(beta), ops._p(mr[0]), ops._p(mr[1]), ops._p(sums),
               float(m * sync.world_size), ops._p(local), ops._p(dx), ops._p(dgamma), ops._p(dbeta), 0, m, c, cs, p.act, p.slope, st)
        return dx
    g = L.NormGeom(n, hw, c, cs, L.NORM_BATCH, p.eps, p.momentum, p.act, p.slope)
    ws = ops.workspace(L.query('cat_norm_ws_bytes', C.byref(g)), x.device)
    L.call('cat_norm_bwd', C.byref(g), ops._p(x), ops._p(dy), ops._p(gamma), ops._p(beta), ops._p(mr[0]), ops._p(mr[1]), ops._p(dx), ops._p(dgamma),
           ops._p(dbeta), 0, ops._p(ws), ops._stream())
    return dx


def _channel_sum(src, m_pix, c, cs, dst):
    ws = ops.workspace(L.query('cat_channel_sum_ws_bytes', m_pix, cs), src.device)
    L.call('cat_channel_sum', ops._p(src), m_pix, c, cs, ops._p(dst), 0, ops._p(ws), ops._stream())


class _UnitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, addend, plan, *params):
        x = ops.conform(x)
        if addend is not None:
            addend = ops.conform(addend)
        save = {}
        y = _drive(forward_g(plan, x, addend, save))
        ctx.plan = plan
        ctx.synced = ops.bn_sync() is not None      # statistics over all ranks: (a | b) saved instead of (mean | rstd)
        ctx.has_dw = save['zd'] is not None
        ctx.has_add = addend is not None
        tensors = [x, save['z1'], save['st1'][0], save['st1'][1]]
        if ctx.has_dw:
            tensors += [save['zd'], save['std'][0], save['std'][1]]
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, dy):
        return _drive(_backward_g(ctx, dy))


def _backward_g(ctx, dy):
    """A unit's backward as a generator over its statistics exchanges.  `ctx`: anything with plan / synced / has_dw / has_add / saved_tensors /
    needs_input_grad (the autograd context of _UnitFn, or the per-unit record of _PrepassFn)."""
    p = ctx.plan
    STATS['bwd'] += 1
    saved = ctx.saved_tensors
    x, z1, ss1, mr1 = saved[:4]
    zd, ssd, mrd = saved[4:] if ctx.has_dw else (None, None, None)
    dt = ops.conform(dy)            # no closing norm: the gradient of the branch sum IS dy (and so is the addend's)
    if ops.act_cs(dt) != p.cso:
        raise RuntimeError('fused SPADE unit backward: gradient pixel stride differs from the activation')
    p.prepare(backward=True)
    n, c, h, w = x.shape
    dev, hw, m_pix = x.device, h * w, n * h * w
    st = ops._stream()
    grads = {}
    side = ops.SideJobs(dev)

    def put_side(param, kernel):
        def job():
            grads[id(param)] = ops._write_param_grad(param, lambda dst_, acc: kernel(dst_, acc, ops._stream()))
        side.run(job)

    # ---- re-materialise the hidden activations (inputs of the second convs / of the depthwise convs)
    a1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
    L.call('cat_affine_res_fwd', ops._p(z1), p.hc1, ops._p(ss1[0]), ops._p(ss1[1]), 0, None, 0, ops._p(a1), p.hc1, 1, m_pix, p.hc1, p.act, p.slope, st)
    da1 = torch.empty((n, h, w, p.hc1), device=dev, dtype=torch.float32)
    ad = dad = None
    if ctx.has_dw:
        ad = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
        L.call('cat_affine_res_fwd', ops._p(zd), p.hcd, ops._p(ssd[0]), ops._p(ssd[1]), 0, None, 0, ops._p(ad), p.hcd, 1, m_pix, p.hcd, p.act, p.slope, st)
        dad = torch.empty((n, h, w, p.hcd), device=dev, dtype=torch.float32)
    # ---- second convs: weight gradients from (hidden activation slice, dT); input gradients into slices of dA1 / dAd
    side.fork()
    for b in p.branches:
        res = b['kind'] == 'res'
        k2, m, w1 = b['k2'], b['m'], b['w1']
        pad2 = (k2 - 1) // 2
        src, scs_, o = (a1, p.hc1, b['o1']) if res else (ad, p.hcd, b['od'])
        dst, dcs = (da1, p.hc1) if res else (dad, p.hcd)
        xptr = C.c_void_p(src.data_ptr() + 4 * o)

        def kw(dst_, acc, sst, xptr=xptr, m=m, scs_=scs_, k2=k2, pad2=pad2):
            gw = ops._conv_geom(n, h, w, m, scs_, h, w, p.Cout, p.cso, k2, k2, 1, pad2, L.PAD_ZERO, wcs=ops._grad_wcs(dst_))
            ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
            L.call('cat_conv2d_wgrad', C.byref(gw), xptr, ops._p(dt), ops._p(dst_), acc, ops._p(ws), sst)
        if res or not p.merge2:
            put_side(b['conv2'].weight, kw)
        if p.s1d is not None and (b is p.s1d[0] or b is p.s1d[1] or not res):
            continue            # input gradient: the merged launch below
        seg = tconv.Segment(None, k2, k2 - 1 - pad2, False, b['d2off'], c4=p.cso, cin=p.Cout, xcs=p.cso, ptr=dt.data_ptr())
        tconv.run([seg], p.dpack2, None, None, m, n, h, w, h, w, ycs=dcs, ycw=w1, yptr=dst.data_ptr() + 4 * o)
    if p.s1d is not None:
        r5, r3 = p.s1d
        gs = L.Stage1Geom()
        gs.N, gs.H, gs.W, gs.xcs, gs.cin, gs.reflect, gs.ycs, gs.scs = n, h, w, p.cso, p.Cout, 0, 0, 0
        packs, dxs, dxcs = (C.c_void_p * 3)(), (C.c_void_p * 3)(), (C.c_int * 3)(p.hc1, p.hc1, p.hcd)
        for slot, (col0, width, nvalid, pk, dst) in enumerate(((r5['o1'], r5['w1'], r5['m'], p.dpack2.data_ptr() + 4 * r5['d2off'], da1),
                                                               (r3['o1'], r3['w1'], r3['m'], p.dpack2.data_ptr() + 4 * r3['d2off'], da1),
                                                               (0, p.hcd, sum(b['m'] for b in p.dws), p.dpack2_dw.data_ptr(), dad))):
            gs.col0[slot], gs.width[slot], gs.nvalid[slot] = col0, width, nvalid
            packs[slot], dxs[slot] = pk, dst.data_ptr()
        L.call('cat_tstage1_dgrad', C.byref(gs), ops._p(dt), packs, dxs, dxcs, ops._stream())
    if p.merge2:
        def kw2(sst):
            gw = ops._conv_geom(n, h, w, p.hcd, p.hcd, h, w, p.Cout, p.cso, 1, 1, 1, 0, L.PAD_ZERO, wcs=p.hcd)
            ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
            L.call('cat_conv2d_wgrad', C.byref(gw), ops._p(ad), ops._p(dt), ops._p(p.gv['w2']), 0, ops._p(ws), sst)
        side.run(lambda: kw2(ops._stream()))
    if p.has_bias2:
        _channel_sum(dt, m_pix, p.Cout, p.cso, p.gv['c2'])
    # ---- depthwise stage
    if ctx.has_dw:
        dzd = yield from _norm_bwd_g(p, n, hw, p.hcd, p.hcd, zd, dad, p.gammad, p.betad, mrd, p.gv['gd'], p.gv['bd'], ctx.synced)
        if p.has_biasd:
            _channel_sum(dzd, m_pix, p.hcd, p.hcd, p.gv['cd'])
        nb = len(p.dws)
        gd = L.DwmGeom()
        gd.N, gd.H, gd.W, gd.nq, gd.xcs, gd.ycs, gd.scs = n, h, w, p.hcd // 4, p.hc1, p.hcd, p.hcd
        gd.reflect = 0
        for b in p.dws:
            for q in range(b['od'] // 4, (b['od'] + _cs4(b['m'])) // 4):
                gd.ks[q] = b['kd']
        wts = [b['dconv'].weight for b in p.dws]
        tg = [ops._grad_target(q) for q in wts]
        if all(t_ is not None for t_ in tg):
            fresh = {q._cat_grad_state['fresh'] for q in wts}
            if len(fresh) != 1:
                raise RuntimeError('fused SPADE unit backward: depthwise gradient buffers out of sync')
            acc_dw, dsts = (0 if fresh.pop() else 1), tg
            for q in wts:
                q._cat_grad_state['fresh'] = False
                grads[id(q)] = None
        else:
            acc_dw, dsts = 0, [torch.empty_like(q) for q in wts]
            for q, d_ in zip(wts, dsts):
                if ops._grad_target(q) is None:
                    grads[id(q)] = d_
        IA = C.c_int * nb
        wsd = ops.workspace(L.query('cat_dwm_bwd_ws_bytes', C.byref(gd)), dev)
        L.call('cat_dwm_bwd', C.byref(gd), C.c_void_p(a1.data_ptr() + 4 * p.dw_in0), ops._p(dzd), ops._p(p.w25),
               C.c_void_p(da1.data_ptr() + 4 * p.dw_in0), p.hc1, nb, IA(*[b['od'] for b in p.dws]), IA(*[b['m'] for b in p.dws]),
               IA(*[b['kd'] for b in p.dws]), (C.c_void_p * nb)(*[d_.data_ptr() for d_ in dsts]), acc_dw, ops._p(wsd), st)
        if not all(t_ is not None for t_ in tg):
            for q, d_ in zip(wts, dsts):
                tq = ops._grad_target(q)
                if tq is not None:
                    (tq.copy_ if q._cat_grad_state['fresh'] else tq.add_)(d_)
                    q._cat_grad_state['fresh'] = False
                    grads[id(q)] = None
    # ---- stage-1 norms (all branches at once)
    dz1 = yield from _norm_bwd_g(p, n, hw, p.hc1, p.hc1, z1, da1, p.gamma1, p.beta1, mr1, p.gv['g1'], p.gv['b1'], ctx.synced)
    if p.has_bias1:
        _channel_sum(dz1, m_pix, p.hc1, p.hc1, p.gv['c1'])
    # ---- first convs: weight gradients from (x, dZ1 slice)
    side.refork()
    if p.merge1 is not None:
        g1 = p.merge1

        def kw1m(sst):
            gw = ops._conv_geom(n, h, w, c, ops.act_cs(x), h, w, g1['width'], p.hc1, 1, 1, 1, 0, L.PAD_ZERO, wcs=p.csi)
            ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
            L.call('cat_conv2d_wgrad', C.byref(gw), ops._p(x), C.c_void_p(dz1.data_ptr() + 4 * g1['off']), ops._p(p.gv['w1']), 0, ops._p(ws), sst)
        side.run(lambda: kw1m(ops._stream()))
    for b in p.branches:
        if p.merge1 is not None and b['k'] == 1:
            continue
        k, m = b['k'], b['m']
        pad1 = (k - 1) // 2
        dyp = C.c_void_p(dz1.data_ptr() + 4 * b['o1'])

        def kw1(dst_, acc, sst, dyp=dyp, m=m, k=k, pad1=pad1):
            gw = ops._conv_geom(n, h, w, c, ops.act_cs(x), h, w, m, p.hc1, k, k, 1, pad1, L.PAD_ZERO, wcs=ops._grad_wcs(dst_))
            ws = ops.workspace(L.query('cat_conv2d_wgrad_ws_bytes', C.byref(gw)), dev)
            L.call('cat_conv2d_wgrad', C.byref(gw), ops._p(x), dyp, ops._p(dst_), acc, ops._p(ws), sst)
        put_side(b['conv1'].weight, kw1)
    # ---- first convs: the input gradients as ONE K-concatenated launch
    dx = None
    if ctx.needs_input_grad[0]:
        segs = []
        for b in p.branches:
            pad1 = (b['k'] - 1) // 2
            segs.append(tconv.Segment(None, b['k'], pad1, False, b['d1off'], c4=b['w1'], cin=b['m'], xcs=p.hc1, ptr=dz1.data_ptr() + 4 * b['o1']))
        dx = ops.empty_act(n, c, h, w, dev)
        tconv.run(segs, p.dpack1, None, dx, c, n, h, w, h, w)
    side.join()
    # ---- scatter the concatenated parameter gradients
    all_t = [q for _, _, _, q in p.targets] + [t2[5] for t2 in p.targets2d]
    owned = [getattr(q, '_cat_grad_view', None) is not None for q in all_t]
    if all_t and all(owned):
        fresh = {q._cat_grad_state['fresh'] for q in all_t}
        if len(fresh) != 1:
            raise RuntimeError('fused SPADE unit backward: gradient buffers of one unit out of sync')
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
        tj, nj, nbk, _ = p.scatter_jobs
        L.call('cat_prep_run', ops._p(tj), nj, nbk, 0 if fresh.pop() else 1, st)
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
            if q.dim() == 4:      # rows of the merged 1 x 1 weight gradient: back into the parameter's [O][1][1][wcs] storage
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
    return (dx, dt if ctx.has_add else None, None) + tuple(grads.get(id(q)) for q in p.params)


class _Rec:
    """Per-unit stand-in for an autograd context inside _PrepassFn.backward (what _backward_g reads)."""
    __slots__ = ('plan', 'synced', 'has_dw', 'has_add', 'saved_tensors', 'needs_input_grad')


class _PrepassFn(torch.autograd.Function):
    """SEVERAL independent units as ONE autograd node (the gamma|beta nets of all SPADE layers of a generator: they read only the segmentation
    map, reference inception_modules.py:746-762).  Forward and backward run the units in lockstep (_drive_many): the k-th SynchronizedBatchNorm
    statistics exchanges of all units are one collective -- two per pass instead of two per unit -- and, being one node, the backward runs when
    the gradients of ALL outputs are there, so its exchanges batch the same way.  Arithmetic per unit is unchanged (bit-identical results)."""

    @staticmethod
    def forward(ctx, plans, nparams, *args):
        n = len(plans)
        xs = [ops.conform(t) for t in args[:n]]
        saves = [dict() for _ in plans]
        ys = _drive_many([forward_g(p, x, None, sv) for p, x, sv in zip(plans, xs, saves)])
        ctx.plans, ctx.nparams = plans, nparams
        ctx.synced = ops.bn_sync() is not None
        ctx.layout, tensors = [], []
        for x, sv in zip(xs, saves):
            t = [x, sv['z1'], sv['st1'][0], sv['st1'][1]]
            if sv['zd'] is not None:
                t += [sv['zd'], sv['std'][0], sv['std'][1]]
            ctx.layout.append(len(t))
            tensors += t
        ctx.save_for_backward(*tensors)
        ctx.set_materialize_grads(False)      # an output nobody used reaches backward as None (no dense NCHW zeros, no layout conversion)
        ctx.out_nhwc = [(y.shape[0], y.shape[2], y.shape[3], ops.act_cs(y), y.shape[1]) for y in ys]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        saved, recs, o = ctx.saved_tensors, [], 0
        for p, cnt in zip(ctx.plans, ctx.layout):
            r = _Rec()
            r.plan, r.synced, r.has_dw, r.has_add = p, ctx.synced, cnt == 7, False
            r.saved_tensors, r.needs_input_grad = saved[o:o + cnt], (False, False)
            o += cnt
            recs.append(r)
        # an output nobody used arrives as None: its unit still has to step through the lockstep exchanges (every rank does the same)
        dys = [dy if dy is not None else torch.zeros(shape[:4], device=saved[0].device, dtype=torch.float32).permute(0, 3, 1, 2)[:, :shape[4]]
               for dy, shape in zip(dys, ctx.out_nhwc)]
        outs = _drive_many([_backward_g(r, dy) for r, dy in zip(recs, dys)])
        grads = []
        for out in outs:
            grads += list(out[3:])
        return (None, None) + (None,) * len(ctx.plans) + tuple(grads)


_PREPASS = os.environ.get('CAT_FUSED_SPADE_PREPASS', '1') != '0'      # A/B switch (round 5)


def set_prepass(on):
    global _PREPASS
    old, _PREPASS = _PREPASS, bool(on)
    return old


def prepass(units):
    """units: [(owner, res_ops, dw_ops, cin, cout, x)] of independent train-mode units whose input needs no gradient (gamma|beta nets on
    the segmentation pyramid).  Returns their outputs, computed in lockstep with merged statistics exchanges, or None if that does not apply
    (no multi-rank reducer, switch off, a unit the fused path does not take): the caller then runs the units one by one as before."""
    if not _PREPASS or ops.bn_sync() is None or not _SYNC_FUSED or _UNITS not in ('all', 'gb') or len(units) < 2:
        return None
    xs = [ops.conform(u[5]) for u in units]
    if any(x.requires_grad for x in xs) or not all(applicable(u[1], u[2], x, True) for u, x in zip(units, xs)):
        return None
    plans = tuple(plan_for(u[0], '_cat_fused_gb', u[1], u[2], u[3], u[4], x) for u, x in zip(units, xs))
    STATS['prepass'] = STATS.get('prepass', 0) + 1
    if not torch.is_grad_enabled():
        return _drive_many([forward_g(p, x, None, None) for p, x in zip(plans, xs)])
    params = [q for p in plans for q in p.params]
    return list(_PrepassFn.apply(plans, tuple(len(p.params) for p in plans), *xs, *params))


def apply(owner, slot, res_ops, dw_ops, cin, cout, x, addend=None):
    """Run the unit `owner`'s branches (res_ops, dw_ops) on x through the fused path; `slot` names the attribute that caches its plan."""
    p = plan_for(owner, slot, res_ops, dw_ops, cin, cout, ops.conform(x))
    if not torch.is_grad_enabled():
        fn = forward if p.branches[0]['bn1'].training else forward_eval
        return fn(p, ops.conform(x), None if addend is None else ops.conform(addend))
    return _UnitFn.apply(x, addend, p, *p.params)
