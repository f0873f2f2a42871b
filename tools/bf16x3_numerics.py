"""Exploratory (round-3 verdict item 9), CPU only: what would split-bf16 arithmetic -- every fp32 operand x = hi + lo with hi = bf16(x),
lo = bf16(x - hi); products hi*hi + hi*lo + lo*hi on the bf16 matrix pipe, fp32 accumulate; the lo*lo term (<= 2^-16 of the product) dropped
-- do to the C2 step if the three wide PatchGAN tiles (conv_fwd32d / conv_dgrad32d / conv_wgrad32d: 48 % of the step at 0.80-0.86 of an fp32
matrix peak that is 1/16 of the bf16 rate) computed that way?  No kernel exists; this restates the arithmetic on the host so that the
decision to build one rests on numbers: the oracle's C2 headline step (256 x 256, batch 2, pruned 4.6e9-MAC student, ndf-128 PatchGAN) is
run in fp64, in fp32, and in fp32 with the discriminator's three 128/256/512-channel 4x4 convolutions -- forward, data gradient and weight
gradient -- replaced by the split form (each bf16 x bf16 product is exact in fp32, so fp32 convolutions of the split operands ARE the
arithmetic of the matrix pipe up to summation order).  Printed in the units of tests/test_spade_gpu.py::check_grads.
    python tools/bf16x3_numerics.py [size] [batch] [terms] [which] [c2 | c3 | c4]      terms: 1 | 2 | 3 (default) | 4 | 6, see combine(); which: subset of 'fdw' (the discriminator's wide layers: forward / dgrad / wgrad; 'F' = forward with 6 terms whatever [terms] says) and 't' (every dense conv of the frozen teacher)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn import grad as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
TERMS = 3
WHICH = 'fdw'      # which of forward / data gradient / weight gradient use the split form
SPADE = False


def split(t, parts=2):
    out, r = [], t
    for _ in range(parts):
        h = r.to(torch.bfloat16).to(torch.float32)      # round to nearest even, as v_cvt_pk_bf16_f32 does
        out.append(h)
        r = r - h
    return out


def combine(fn, a, b, terms=None):
    """fn over operands a, b (linear in both) in split arithmetic.  TERMS = 1: a1 b1 (plain bf16); 2: + a1 b2; 3: + a2 b1 ("bf16x3": what is
    dropped is <= 2^-16 of a product); 4: + a2 b2 (both operands cut to 16 mantissa bits, 2^-17); 6: three-way split, the six products with
    i + j <= 4 (2^-24: fp32 class).  A sum of parts is exact in fp32, and so is an 8-bit x 16-bit product, so a1 * (b1 + b2) is ONE call;
    8 x 24 bits rounds at 2^-24, which is the class the 6-term form is in anyway."""
    terms = TERMS if terms is None else terms
    if terms == 6:
        a1, a2, a3 = split(a, 3)
        b1, b2, b3 = split(b, 3)
        return fn(a1, b1 + b2 + b3) + fn(a2, b1 + b2) + fn(a3, b1)
    a1, a2 = split(a)
    b1, b2 = split(b)
    if terms == 1:
        return fn(a1, b1)
    if terms == 2:
        return fn(a1, b1 + b2)
    if terms == 3:
        return fn(a1, b1 + b2) + fn(a2, b1)
    return fn(a1 + a2, b1 + b2)


class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, which=None):
        which = WHICH if which is None else which
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, padding, which)
        fn = lambda a, b: _conv(a, b, None, stride, padding)
        return combine(fn, x, w, 6) if 'F' in which else (combine(fn, x, w) if 'f' in which else fn(x, w))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, which = ctx.geom
        fd = lambda a, b: G.conv2d_input(x.shape, b, a, stride, padding)
        fw = lambda a, b: G.conv2d_weight(b, w.shape, a, stride, padding)
        dx = (combine(fd, dy, w) if 'd' in which else fd(dy, w)) if ctx.needs_input_grad[0] else None
        dw = (combine(fw, dy, x) if 'w' in which else fw(dy, x)) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


_conv = F.conv2d
STATS = {'calls': 0}


def patched_conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    # the layers the library runs on the three direct-to-LDS wide tiles (csrc/conv_igemm.hip: fwd_bk32_ok / dgrad32d_ok / wgrad32d_nsplit)
    dense = x.dtype == torch.float32 and groups == 1 and dilation in (1, (1, 1))
    if SPADE and dense:      # GauGAN: the eligibility rules themselves (VGG19's and the discriminators' layers; each direction on its own rule)
        co, ci = w.shape[0], w.shape[1]
        el = ('f' if ci % 16 == 0 and ci >= 64 and co >= 32 else '') + ('d' if ci > 96 and co % 32 == 0 else '') + ('w' if co > 96 and ci % 128 == 0 else '')
        use = ''.join(c for c in el if c in WHICH or (c == 'f' and 'F' in WHICH))
        if not use:
            return _conv(x, w, b, stride, padding, dilation, groups)
        STATS['calls'] += 1
        y = SplitConv.apply(x, w, stride, padding, use + ('F' if 'F' in WHICH and 'f' in el else ''))
        return y if b is None else y + b.view(1, -1, 1, 1)
    wide = dense and w.shape[2:] == (4, 4) and w.shape[1] % 128 == 0 and w.shape[0] > 96
    if 't' in WHICH and not torch.is_grad_enabled() and x.dtype == torch.float32 and groups == 1 and w.shape[1] >= 16:
        # the frozen teacher's dense convolutions (distill_step runs it under no_grad): forward only, nothing is differentiated through it
        STATS['teacher'] = STATS.get('teacher', 0) + 1
        y = combine(lambda a, c: _conv(a, c, None, stride, padding, dilation, 1), x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)
    if not wide or not (set(WHICH) & set('fFdw')):
        return _conv(x, w, b, stride, padding, dilation, groups)
    STATS['calls'] += 1
    y = SplitConv.apply(x, w, stride, padding)
    return y if b is None else y + b.view(1, -1, 1, 1)


def report(name, got, ref64, own32=None):
    """check_grads' units: per tensor max |d| / max(own max, 3 % of the global max); the fp32 oracle's own figure beside it"""
    gmax = max(float(v.abs().max()) for v in ref64.values())
    rows = []
    for k, v in ref64.items():
        den = max(float(v.abs().max()), 3e-2 * gmax)
        e = float((got[k].double() - v).abs().max()) / den
        o = float((own32[k].double() - v).abs().max()) / den if own32 is not None else float('nan')
        rows.append((e, o, k))
    e, o = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
    print('%s, %d tensors [split | plain fp32]: median %.2e | %.2e   90 %% quantile %.2e | %.2e   worst %.2e | %.2e   within 1e-3: %.1f %% | %.1f %%' % (
        name, len(rows), np.median(e), np.median(o), np.quantile(e, 0.9), np.quantile(o, 0.9), e.max(), o.max(), 100 * (e < 1e-3).mean(), 100 * (o < 1e-3).mean()))
    net = [r for r in rows if r[0] > max(10 * r[1], 5e-3)]
    print('    per-tensor net  split <= max(10 x fp32, 5e-3): %d fail %s' % (len(net), [(r[2], float('%.2e' % r[0]), float('%.2e' % r[1])) for r in sorted(net, reverse=True)[:4]]))
    return rows


def main():
    global TERMS, WHICH, SPADE
    from oracle import detfill, ref_cpu
    from oracle_fp64_calibration import oracle_cfg, state_dicts, to64
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    TERMS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    WHICH = sys.argv[4] if len(sys.argv) > 4 else 'fdw'
    which = sys.argv[5] if len(sys.argv) > 5 else 'c2'
    if which == 'c4':
        return main_spade(nb)
    opt, T, S, D = state_dicts(which)
    cfg = oracle_cfg(opt)
    A, B = detfill.images((nb, 3, size, size), 71), detfill.images((nb, 3, size, size), 72)
    st64 = ref_cpu.DistillState(to64(T), to64(S), to64(D), cfg)
    ref_cpu.distill_step(st64, A.double(), B.double())
    st32 = ref_cpu.DistillState(T, S, D, cfg)
    ref_cpu.distill_step(st32, A, B)
    stx = ref_cpu.DistillState(T, S, D, cfg)
    ref_cpu.F.conv2d = patched_conv2d
    try:
        ref_cpu.distill_step(stx, A, B)
    finally:
        ref_cpu.F.conv2d = _conv
    print(which.upper() + ' step @%dx%d batch %d; %d discriminator convolutions (x3: forward, dgrad, wgrad) in %d-term split-bf16 arithmetic, applied to [%s]; teacher convolutions in split form: %d' % (size, size, nb, STATS['calls'], TERMS, WHICH, STATS.get('teacher', 0)))
    for k in st64.losses:
        r = abs(st64.losses[k]) + 1e-30
        print('  loss %-11s fp64 %+.8f   split: %.2e   plain fp32: %.2e   (relative)' % (k, st64.losses[k], abs(stx.losses[k] - st64.losses[k]) / r, abs(st32.losses[k] - st64.losses[k]) / r))
    for k in ('Sfake_B', 'Tfake_B'):
        r64 = getattr(st64, k)
        print('  image %-8s max |d| / max |ref|:  split %.2e   plain fp32 %.2e' % (k, float((getattr(stx, k).double() - r64).abs().max() / r64.abs().max()),
                                                                                float((getattr(st32, k).double() - r64).abs().max() / r64.abs().max())))
    report('  student gradients', stx.grads_S, st64.grads_S, st32.grads_S)
    report('  discriminator gradients', stx.grads_D, st64.grads_D, st32.grads_D)


def main_spade(nb):
    """The GauGAN step (BASELINE configs[3], 512 x 256): VGG19's and the two discriminators' wide layers in split form."""
    global SPADE
    from oracle import detfill, ref_spade_cpu as R
    from oracle_fp64_calibration import spade_state_dicts, to64
    SPADE = True
    opt, cfg, T, S, D, V = spade_state_dicts()
    h, w = 256, 512
    rng = np.random.default_rng(5)
    lab = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 35, (nb, 1, h // 16, w // 16)), 16, 2), 16, 3))
    ins = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 1000, (nb, 1, h // 16, w // 16)), 16, 2), 16, 3).astype(np.int32))
    img = detfill.images((nb, 3, h, w), 6)
    sem = R.preprocess_input(lab, ins, 35)
    st64 = R.SpadeState(to64(T), to64(S), to64(D), to64(V), cfg)
    R.spade_step(st64, sem.double(), img.double())
    st32 = R.SpadeState(T, S, D, V, cfg)
    R.spade_step(st32, sem, img)
    stx = R.SpadeState(T, S, D, V, cfg)
    F.conv2d = patched_conv2d
    try:
        R.spade_step(stx, sem, img)
    finally:
        F.conv2d = _conv
    print('C4 (GauGAN) step @512x256 batch %d; %d convolution calls in %d-term split-bf16 arithmetic, directions [%s] where the layer is eligible' % (nb, STATS['calls'], TERMS, WHICH))
    for k in st64.losses:
        r = max(abs(st64.losses[k]), 1e-2)
        print('  loss %-11s fp64 %+.8f   split: %.2e   plain fp32: %.2e   (relative)' % (k, st64.losses[k], abs(stx.losses[k] - st64.losses[k]) / r, abs(st32.losses[k] - st64.losses[k]) / r))
    for k in ('Sfake_B', 'Tfake_B'):
        r64 = getattr(st64, k)
        print('  image %-8s max |d| / max |ref|:  split %.2e   plain fp32 %.2e' % (k, float((getattr(stx, k).double() - r64).abs().max() / r64.abs().max()),
                                                                                float((getattr(st32, k).double() - r64).abs().max() / r64.abs().max())))
    report('  student gradients', stx.grads_S, st64.grads_S, st32.grads_S)
    report('  discriminator gradients', stx.grads_D, st64.grads_D, st32.grads_D)


if __name__ == '__main__':
    main()
