"""Host-side emulation of conv_dgrad32_kernel's data movement (global -> registers -> swizzled LDS -> MFMA fragments) in numpy.

Mirrors the kernel statement by statement (walk state, parked pointers, swizzles, fragment reads) for one launch and compares
the result with a direct dgrad.  Purpose: catch indexing mistakes on a machine without a GPU.  Not part of the product."""
import sys

import numpy as np

MT, NT, WM, WN = 4, 4, 2, 2
BM, BN = WM * MT * 16, WN * NT * 16
AI, BQ = BM // 32, BN // 4
BR = 256 // BQ
BI = 32 // BR


def swz32(r, q):
    return (r * 8 + (q ^ ((r >> 1) & 7))) * 4


def run(N, H, W, Cin, Cout, k, stride, pad, seed=0):
    rng = np.random.default_rng(seed)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    ycs, wcs, ocs = Cout, Cin, Cin
    dy = rng.standard_normal((N, Ho, Wo, ycs)).astype(np.float64)
    w = rng.standard_normal((Cout, k, k, wcs)).astype(np.float64)
    # reference: dx[n, iy, ix, ci] = sum dy[n, oy, ox, co] * w[co, ky, kx, ci] with iy = oy*s - pad + ky
    ref = np.zeros((N, H, W, Cin))
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(k):
                for kx in range(k):
                    iy, ix = oy * stride - pad + ky, ox * stride - pad + kx
                    if 0 <= iy < H and 0 <= ix < W:
                        ref[:, iy, ix, :] += dy[:, oy, ox, :] @ w[:, ky, kx, :]
    out = np.full((N, H, W, ocs), np.nan)
    dyf, wf = dy.reshape(-1), w.reshape(-1)
    c4 = Cout
    Hin, Win, pad_eff, s = H, W, pad, stride
    taps = k * k
    mmax = N * (-(-Hin // s)) * (-(-Win // s))
    gx = (-(-mmax // BM)) * (-(-Cin // BN))
    for by in range(s * s):
        py, px = by // s, by % s
        iyf, ixf = ((py - pad_eff) % s + s) % s, ((px - pad_eff) % s + s) % s
        Hc = (Hin - iyf + s - 1) // s if iyf < Hin else 0
        Wc = (Win - ixf + s - 1) // s if ixf < Win else 0
        cy0, cx0 = (iyf + pad_eff - py) // s, (ixf + pad_eff - px) // s
        nty = (k - py + s - 1) // s if py < k else 0
        ntx = (k - px + s - 1) // s if px < k else 0
        ntaps, ntxd = nty * ntx, max(ntx, 1)
        K = ntaps * c4
        Mc = N * Hc * Wc
        ntn = -(-Cin // BN)
        for bid in range(gx):      # xcd_remap is a bijection: the order does not matter here
            m0, n0 = (bid // ntn) * BM, (bid % ntn) * BN
            if m0 >= Mc:
                continue
            HcWc = Hc * Wc
            # per-thread state
            T = []
            for tid in range(256):
                wave = tid >> 6
                half = wave >> 1
                q4 = tid & 3
                q = half * 4 + q4
                r0 = (tid & 127) >> 2
                st = dict(q=q, r0=r0, cy=[], cx=[], aoff=[], rv=[])
                for i in range(AI):
                    m = m0 + r0 + 32 * i
                    rv = m < Mc
                    mm = m if rv else 0
                    n, rem = divmod(mm, HcWc)
                    a, b = divmod(rem, Wc)
                    st['cy'].append(cy0 + a); st['cx'].append(cx0 + b); st['aoff'].append(n * Ho * Wo * ycs); st['rv'].append(rv)
                ka = q * 4
                st['atj'] = ka // c4
                st['aco'] = ka - st['atj'] * c4
                st['ajy'] = st['atj'] // ntxd
                st['ajx'] = st['atj'] - st['ajy'] * ntxd
                st['nq'] = tid % BQ
                st['bci'] = n0 + st['nq'] * 4
                st['bcv'] = st['bci'] < Cin
                st['btj'], st['bco'], st['kr'] = [], [], []
                for i in range(BI):
                    kr = tid // BQ + BR * i
                    st['kr'].append(kr)
                    st['btj'].append(kr // c4)
                    st['bco'].append(kr - (kr // c4) * c4)
                T.append(st)

            def locate_a(st):
                tv = st['atj'] < ntaps
                st['pa'], st['inca'] = [], []
                for i in range(AI):
                    oy, ox = st['cy'][i] - st['ajy'], st['cx'][i] - st['ajx']
                    v = st['rv'][i] and tv and 0 <= oy < Ho and 0 <= ox < Wo
                    st['pa'].append(st['aoff'][i] + (oy * Wo + ox) * ycs + st['aco'] if v else None)
                    st['inca'].append(32 if v else 0)

            def locate_b(st, i):
                jy = st['btj'][i] // ntxd
                jx = st['btj'][i] - jy * ntxd
                ky, kx = py + jy * s, px + jx * s
                v = st['bcv'] and st['btj'][i] < ntaps
                st['pb'][i] = (st['bco'][i] * taps + ky * k + kx) * wcs + st['bci'] if v else None
                st['incb'][i] = 32 * taps * wcs if v else 0

            for st in T:
                locate_a(st)
                st['pb'], st['incb'] = [None] * BI, [0] * BI
                for i in range(BI):
                    locate_b(st, i)

            def gload():
                for st in T:
                    st['ra'] = [dyf[p:p + 4].copy() if p is not None else np.zeros(4) for p in st['pa']]
                    st['rb'] = [wf[p:p + 4].copy() if p is not None else np.zeros(4) for p in st['pb']]
                    st['aco'] += 32
                    if st['aco'] >= c4:
                        while True:
                            st['aco'] -= c4
                            st['atj'] += 1
                            st['ajx'] += 1
                            if st['ajx'] == ntxd:
                                st['ajx'] = 0
                                st['ajy'] += 1
                            if st['aco'] < c4:
                                break
                        locate_a(st)
                    else:
                        st['pa'] = [p + inc if p is not None else None for p, inc in zip(st['pa'], st['inca'])]
                    for i in range(BI):
                        st['bco'][i] += 32
                        if st['bco'][i] >= c4:
                            while True:
                                st['bco'][i] -= c4
                                st['btj'][i] += 1
                                if st['bco'][i] < c4:
                                    break
                            locate_b(st, i)
                        elif st['pb'][i] is not None:
                            st['pb'][i] += st['incb'][i]

            sA = np.full((2, BM * 32), np.nan)
            sB = np.full((2, 32 * BN), np.nan)

            def sstore(buf):
                for st in T:
                    for i in range(AI):
                        o = swz32(st['r0'] + 32 * i, st['q'])
                        sA[buf, o:o + 4] = st['ra'][i]
                    for i in range(BI):
                        kr = st['kr'][i]
                        o = kr * BN + ((st['nq'] * 4) ^ (((kr >> 2) & 3) << 4))
                        sB[buf, o:o + 4] = st['rb'][i]

            acc = np.zeros((4, MT, NT, 16, 16))   # per wave: D tiles

            def mma(buf):
                A, B = sA[buf], sB[buf]
                for wave in range(4):
                    wm, wn = wave // WN, wave % WN
                    for h in range(2):
                        for t in range(4):
                            for i in range(MT):
                                A4 = np.zeros((16, 4))
                                for lr in range(16):
                                    for lq in range(4):
                                        A4[lr, lq] = A[swz32(wm * MT * 16 + i * 16 + lr, lq + 4 * h) + t]
                                for j in range(NT):
                                    B4 = np.zeros((4, 16))
                                    for lr in range(16):
                                        for lq in range(4):
                                            B4[lq, lr] = B[(h * 16 + lq * 4 + t) * BN + ((wn * NT * 16 + j * 16 + lr) ^ (lq << 4))]
                                    acc[wave, i, j] += A4 @ B4

            nk = (K + 31) >> 5
            if nk > 0:
                gload(); sstore(0)
                for kc in range(nk - 1):
                    buf = kc & 1
                    gload(); mma(buf); sstore(buf ^ 1)
                mma((nk - 1) & 1)
            for wave in range(4):
                wm, wn = wave // WN, wave % WN
                for i in range(MT):
                    for row in range(16):
                        m = m0 + wm * MT * 16 + i * 16 + row
                        if m >= Mc:
                            continue
                        n, rem = divmod(m, HcWc)
                        a, b = divmod(rem, Wc)
                        for j in range(NT):
                            for lr in range(16):
                                col = n0 + wn * NT * 16 + j * 16 + lr
                                if col < Cin:
                                    out[n, iyf + a * s, ixf + b * s, col] = acc[wave, i, j, row, lr]
    assert not np.isnan(out).any(), 'unwritten outputs'
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print(f'N{N} {H}x{W} Cin{Cin} Cout{Cout} k{k} s{stride} p{pad}: rel err {err:.2e}')
    assert err < 1e-12


if __name__ == '__main__':
    run(1, 5, 6, 132, 48, 3, 1, 1)
    run(2, 8, 8, 128, 16, 4, 2, 1)
    run(1, 7, 5, 136, 32, 3, 2, 1)
    run(1, 6, 6, 128, 64, 1, 1, 0)
    print('ok')
