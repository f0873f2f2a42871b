"""Host-side emulation of cat_tile::fwd_kernel (conv_tile.hip) in numpy: staging map, [row][col][16+4] LDS tile, MFMA fragment
addressing, epilogue -- statement by statement, against a direct convolution.  Not part of the product."""
import numpy as np

TH, TW, PITCH, CK = 8, 32, 20, 16


def reflect_idx(i, n):
    i = -i if i < 0 else i
    return 2 * n - 2 - i if i >= n else i


def core(x, w_flat, bias, N, H, W, xcs, Ho, Wo, ycs, Ck, Nn, cw, padv, reflect, KS, wcs, dgrad):
    """cat_tile::tile_kernel, one launch.  x: [N,H,W,xcs] staged operand; returns y [N,Ho,Wo,ycs] (NaN where nothing is written)."""
    NT = -(-Nn // 16)
    c4 = (Ck + 3) & ~3
    y = np.full((N, Ho, Wo, ycs), np.nan)
    xf = x.reshape(-1)
    TR, TC, TAPS = TH + KS - 1, TW + KS - 1, KS * KS
    SLOTS = TR * TC * 4
    ITERS = (SLOTS + 255) // 256
    tiles_x, tiles_y = -(-Wo // TW), -(-Ho // TH)
    for n in range(N):
        for bx in range(tiles_x * tiles_y):
            oy0, ox0 = (bx // tiles_x) * TH, (bx % tiles_x) * TW
            soff, sval = {}, {}
            for tid in range(256):
                for it in range(ITERS):
                    idx = tid + it * 256
                    pix, quad = idx >> 2, idx & 3
                    r, c = divmod(pix, TC)
                    iy, ix = oy0 - padv + r, ox0 - padv + c
                    v = idx < SLOTS
                    if reflect:
                        v = v and -padv <= iy < H + padv and -padv <= ix < W + padv
                        iy, ix = reflect_idx(iy, H), reflect_idx(ix, W)
                    else:
                        v = v and 0 <= iy < H and 0 <= ix < W
                    sval[tid, it] = v
                    soff[tid, it] = ((n * H + iy) * W + ix) * xcs + quad * 4 if v else 0
            tile = np.full(TR * TC * PITCH, np.nan)

            def gload(c0):
                regs = {}
                for tid in range(256):
                    for it in range(ITERS):
                        quad = (tid + it * 256) & 3
                        v = sval[tid, it] and c0 + quad * 4 < c4
                        o = soff[tid, it] + c0
                        regs[tid, it] = xf[o:o + 4].copy() if v else np.zeros(4)
                return regs

            def sstore(regs):
                for tid in range(256):
                    for it in range(ITERS):
                        idx = tid + it * 256
                        if idx < SLOTS:
                            o = (idx >> 2) * PITCH + (idx & 3) * 4
                            tile[o:o + 4] = regs[tid, it]

            acc = np.zeros((4, 4, NT, 16, 16))
            nch = -(-c4 // CK)
            regs = gload(0)
            for ch in range(nch):
                c0 = ch * CK
                sstore(regs)
                if ch + 1 < nch:
                    regs = gload(c0 + CK)
                # B operand per (tap, j): B4[lq][lr] per t, following the kernel's pointer walk
                for tap in range(TAPS):
                    ky, kx = divmod(tap, KS)
                    Bt = np.zeros((NT, 4, 4, 16))     # j, t, lq, lr
                    for j in range(NT):
                        for lr in range(16):
                            nn = j * 16 + lr
                            bval = nn < Nn
                            for lq in range(4):
                                for t in range(4):
                                    if dgrad:
                                        v = bval and c0 + lq * 4 + t < Ck
                                        if v:
                                            brow = ((lq * 4) * TAPS + (TAPS - 1)) * wcs + nn
                                            Bt[j, t, lq, lr] = w_flat[brow + (c0 + t) * TAPS * wcs - tap * wcs]
                                    else:
                                        v = bval and c0 + lq * 4 < c4
                                        if v:
                                            brow = nn * TAPS * wcs + lq * 4
                                            Bt[j, t, lq, lr] = w_flat[brow + c0 + tap * wcs + t]
                    for wave in range(4):
                        for t in range(4):
                            for i in range(4):
                                A4 = np.zeros((16, 4))
                                for lr in range(16):
                                    for lq in range(4):
                                        abase = ((2 * wave) * TC + lr) * PITCH + lq * 4
                                        A4[lr, lq] = tile[abase + (((i >> 1) + ky) * TC + (i & 1) * 16 + kx) * PITCH + t]
                                for j in range(NT):
                                    acc[wave, i, j] += A4 @ Bt[j, t]
            for wave in range(4):
                for i in range(4):
                    oy = oy0 + 2 * wave + (i >> 1)
                    if oy >= Ho:
                        continue
                    for row in range(16):
                        ox = ox0 + (i & 1) * 16 + row
                        if ox >= Wo:
                            continue
                        for j in range(NT):
                            for lr in range(16):
                                co = j * 16 + lr
                                if co < Nn:
                                    y[n, oy, ox, co] = acc[wave, i, j, row, lr] + (bias[co] if bias is not None else 0.0)
                                elif co < cw:
                                    y[n, oy, ox, co] = 0.0
    return y


def run_dgrad(N, H, W, Cin, Cout, KS, reflect, seed=0):
    """dgrad of a stride-1 'same' conv through the tile kernel's DGRAD convention (conv_tile_dgrad)."""
    rng = np.random.default_rng(seed)
    pad = (KS - 1) // 2
    wcs = (Cin + 3) & ~3
    ycs = (Cout + 3) & ~3
    dy = np.zeros((N, H, W, ycs)); dy[..., :Cout] = rng.standard_normal((N, H, W, Cout))
    w = np.zeros((Cout, KS, KS, wcs)); w[..., :Cin] = rng.standard_normal((Cout, KS, KS, Cin))
    Hin, Win, pad_eff = (H + 2 * pad, W + 2 * pad, 0) if reflect else (H, W, pad)
    ref = np.zeros((N, Hin, Win, Cin))     # gradient w.r.t. the (padded, if reflect) input plane
    for oy in range(H):
        for ox in range(W):
            for ky in range(KS):
                for kx in range(KS):
                    iy, ix = oy - pad_eff + ky, ox - pad_eff + kx
                    if 0 <= iy < Hin and 0 <= ix < Win:
                        ref[:, iy, ix, :] += dy[:, oy, ox, :Cout] @ w[:, ky, kx, :Cin]
    dxcs = (Cin + 3) & ~3
    out = core(dy, w.reshape(-1), None, N, H, W, ycs, Hin, Win, dxcs, Cout, Cin, dxcs, KS - 1 - pad_eff, False, KS, wcs, True)
    assert not np.isnan(out).any() and (out[..., Cin:] == 0).all()
    err = np.abs(out[..., :Cin] - ref).max() / np.abs(ref).max()
    print(f'dgrad N{N} {H}x{W} Cin{Cin} Cout{Cout} k{KS} reflect{reflect}: rel err {err:.2e}')
    assert err < 1e-12


def run(N, H, W, Cin, Cout, KS, reflect, seed=0, ycw=0):
    rng = np.random.default_rng(seed)
    pad = (KS - 1) // 2
    NT = -(-Cout // 16)
    c4 = (Cin + 3) & ~3
    xcs, wcs, ycs = c4, c4, (max(Cout, ycw) + 3) & ~3
    cw = max(ycw, Cout)
    assert cw <= NT * 16
    x = np.zeros((N, H, W, xcs)); x[..., :Cin] = rng.standard_normal((N, H, W, Cin))
    w = np.zeros((Cout, KS, KS, wcs)); w[..., :Cin] = rng.standard_normal((Cout, KS, KS, Cin))
    bias = rng.standard_normal(Cout)
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)), mode='reflect' if reflect else 'constant')
    ref = np.zeros((N, H, W, Cout))
    for ky in range(KS):
        for kx in range(KS):
            ref += xp[:, ky:ky + H, kx:kx + W, :] @ w[:, ky, kx, :].T
    ref += bias
    y = core(x, w.reshape(-1), bias, N, H, W, xcs, H, W, ycs, Cin, Cout, cw, pad, reflect, KS, wcs, False)
    assert not np.isnan(y[..., :cw]).any(), 'unwritten outputs'
    assert (y[..., Cout:cw] == 0).all()
    err = np.abs(y[..., :Cout] - ref).max() / np.abs(ref).max()
    print(f'N{N} {H}x{W} Cin{Cin} Cout{Cout} k{KS} reflect{reflect}: rel err {err:.2e}')
    assert err < 1e-12


if __name__ == '__main__':
    run(1, 9, 35, 22, 18, 5, True, ycw=20)
    run(2, 8, 32, 10, 7, 3, False, ycw=8)
    run(1, 11, 40, 36, 42, 5, False, ycw=44)
    run(1, 16, 33, 16, 16, 3, True)
    run_dgrad(1, 9, 33, 18, 82, 5, True)
    run_dgrad(2, 8, 20, 12, 22, 3, False)
    run_dgrad(1, 10, 34, 17, 16, 5, False)
    print('ok')


def run_wgrad(N, H, W, Cin, Cout, KS, reflect, nz, seed=0):
    """cat_tile::wgrad_tile_kernel + the partial reduce, against a direct weight gradient."""
    rng = np.random.default_rng(seed)
    pad = (KS - 1) // 2
    NCO = 1 if Cout <= 16 else 2
    c4 = (Cin + 3) & ~3
    xcs, ycs = c4, (Cout + 3) & ~3
    cout4 = ycs
    x = np.zeros((N, H, W, xcs)); x[..., :Cin] = rng.standard_normal((N, H, W, Cin))
    dy = np.zeros((N, H, W, ycs)); dy[..., :Cout] = rng.standard_normal((N, H, W, Cout))
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)), mode='reflect' if reflect else 'constant')
    TAPS = KS * KS
    K = TAPS * c4
    ref = np.zeros((Cout, TAPS, c4))
    for ky in range(KS):
        for kx in range(KS):
            ref[:, ky * KS + kx, :] = np.einsum('nhwo,nhwc->oc', dy[..., :Cout], xp[:, ky:ky + H, kx:kx + W, :])
    TR, TC = TH + KS - 1, TW + KS - 1
    XSLOTS = TR * TC * 4
    XITERS = (XSLOTS + 255) // 256
    DSLOTS = TH * TW * 4 * NCO
    DITERS = DSLOTS // 256
    NSLOT = (TAPS + 3) // 4
    tiles_x, tiles_y = -(-W // TW), -(-H // TH)
    ntiles = N * tiles_x * tiles_y
    part = np.full((nz, Cout, K), np.nan)
    xf, dyf = x.reshape(-1), dy.reshape(-1)
    for bz in range(-(-c4 // 16)):
        for by in range(-(-Cout // (16 * NCO))):
            for bx in range(nz):
                co0, c0 = by * 16 * NCO, bz * 16
                acc = np.zeros((4, NSLOT, NCO, 16, 16))     # wave, slot, co tile, m = co, n = ci
                for t in range(bx, ntiles, nz):
                    per = tiles_x * tiles_y
                    n, tt = divmod(t, per)
                    oy0, ox0 = (tt // tiles_x) * TH, (tt % tiles_x) * TW
                    xt = np.full(TR * TC * 16, np.nan)
                    dyt = np.full(NCO * TH * TW * 16, np.nan)
                    for tid in range(256):
                        for it in range(XITERS):
                            idx = tid + it * 256
                            pix, quad = idx >> 2, idx & 3
                            r, c = divmod(pix, TC)
                            iy, ix = oy0 - pad + r, ox0 - pad + c
                            v = idx < XSLOTS and c0 + quad * 4 < c4
                            if reflect:
                                v = v and -pad <= iy < H + pad and -pad <= ix < W + pad
                                iy, ix = reflect_idx(iy, H), reflect_idx(ix, W)
                            else:
                                v = v and 0 <= iy < H and 0 <= ix < W
                            o = ((n * H + iy) * W + ix) * xcs + c0 + quad * 4
                            val = xf[o:o + 4] if v else np.zeros(4)
                            if idx < XSLOTS:
                                xt[idx * 4:idx * 4 + 4] = val
                        for it in range(DITERS):
                            idx = tid + it * 256
                            pix, quad = divmod(idx, 4 * NCO)
                            oy, ox = oy0 + pix // TW, ox0 + pix % TW
                            v = oy < H and ox < W and co0 + quad * 4 < cout4
                            o = ((n * H + oy) * W + ox) * ycs + co0 + quad * 4
                            val = dyf[o:o + 4] if v else np.zeros(4)
                            d = ((quad >> 2) * TH * TW + pix) * 16 + (quad & 3) * 4
                            dyt[d:d + 4] = val
                    for wave in range(4):
                        for r in range(TH):
                            for cg in range(TW // 4):
                                A = np.zeros((NCO, 16, 4))      # [co tile][m = co (lr)][k = pixel (lq)]
                                for j in range(NCO):
                                    for lr in range(16):
                                        for lq in range(4):
                                            A[j, lr, lq] = dyt[(j * TH * TW + r * TW + cg * 4 + lq) * 16 + lr]
                                for s in range(NSLOT):
                                    if s + 1 < NSLOT or wave == 0:
                                        tap = min(wave + 4 * s, TAPS - 1)
                                        toff = ((tap // KS) * TC + tap % KS) * 16
                                        B = np.zeros((4, 16))
                                        for lr in range(16):
                                            for lq in range(4):
                                                B[lq, lr] = xt[toff + (r * TC + cg * 4 + lq) * 16 + lr]
                                        for j in range(NCO):
                                            acc[wave, s, j] += A[j] @ B
                for wave in range(4):
                    for s in range(NSLOT):
                        tap = wave + 4 * s
                        if tap >= TAPS:
                            continue
                        for lr in range(16):
                            ci = c0 + lr
                            if ci >= c4:
                                continue
                            for j in range(NCO):
                                for row in range(16):
                                    co = co0 + j * 16 + row
                                    if co < Cout:
                                        part[bx, co, tap * c4 + ci] = acc[wave, s, j, row, lr]
    assert not np.isnan(part).any(), 'unwritten partials'
    got = part.sum(0).reshape(Cout, TAPS, c4)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f'wgrad N{N} {H}x{W} Cin{Cin} Cout{Cout} k{KS} reflect{reflect} nz{nz}: rel err {err:.2e}')
    assert err < 1e-12


if __name__ == '__main__':
    run_wgrad(1, 9, 35, 22, 18, 5, True, 2)
    run_wgrad(2, 8, 40, 18, 40, 3, False, 3)
    run_wgrad(1, 16, 32, 6, 7, 5, False, 1)
    print('wgrad ok')
