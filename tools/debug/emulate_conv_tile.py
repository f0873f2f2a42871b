"""Host-side emulation of cat_tile::fwd_kernel (conv_tile.hip) in numpy: staging map, [row][col][16+4] LDS tile, MFMA fragment
addressing, epilogue -- statement by statement, against a direct convolution.  Not part of the product."""
import numpy as np

TH, TW, PITCH, CK = 8, 32, 20, 16


def reflect_idx(i, n):
    i = -i if i < 0 else i
    return 2 * n - 2 - i if i >= n else i


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
    y = np.full((N, H, W, ycs), np.nan)
    xf, wf = x.reshape(-1), w.reshape(-1)
    TR, TC, TAPS = TH + KS - 1, TW + KS - 1, KS * KS
    SLOTS = TR * TC * 4
    ITERS = (SLOTS + 255) // 256
    tiles_x, tiles_y = -(-W // TW), -(-H // TH)
    for n in range(N):
        for bx in range(tiles_x * tiles_y):
            oy0, ox0 = (bx // tiles_x) * TH, (bx % tiles_x) * TW
            soff, sval = {}, {}
            for tid in range(256):
                for it in range(ITERS):
                    idx = tid + it * 256
                    pix, quad = idx >> 2, idx & 3
                    r, c = divmod(pix, TC)
                    iy, ix = oy0 - pad + r, ox0 - pad + c
                    v = idx < SLOTS
                    if reflect:
                        v = v and -pad <= iy < H + pad and -pad <= ix < W + pad
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

            acc = np.zeros((4, 4, NT, 16, 16))   # wave, i, j, row(pixel), col(channel)
            nch = -(-c4 // CK)
            regs = gload(0)
            for ch in range(nch):
                c0 = ch * CK
                sstore(regs)
                if ch + 1 < nch:
                    regs = gload(c0 + CK)
                for wave in range(4):
                    for tap in range(TAPS):
                        ky, kx = divmod(tap, KS)
                        for t in range(4):
                            for i in range(4):
                                A4 = np.zeros((16, 4))
                                for lr in range(16):
                                    for lq in range(4):
                                        abase = ((2 * wave) * TC + lr) * PITCH + lq * 4
                                        A4[lr, lq] = tile[abase + (((i >> 1) + ky) * TC + (i & 1) * 16 + kx) * PITCH + t]
                                for j in range(NT):
                                    B4 = np.zeros((4, 16))
                                    for lr in range(16):
                                        co = j * 16 + lr
                                        for lq in range(4):
                                            v = co < Cout and c0 + lq * 4 < c4
                                            if v:
                                                B4[lq, lr] = wf[(co * TAPS) * wcs + lq * 4 + c0 + tap * wcs + t]
                                    acc[wave, i, j] += A4 @ B4
            for wave in range(4):
                for i in range(4):
                    oy = oy0 + 2 * wave + (i >> 1)
                    if oy >= H:
                        continue
                    for row in range(16):          # row = lq*4 + rg
                        ox = ox0 + (i & 1) * 16 + row
                        if ox >= W:
                            continue
                        for j in range(NT):
                            for lr in range(16):
                                co = j * 16 + lr
                                if co < Cout:
                                    y[n, oy, ox, co] = acc[wave, i, j, row, lr] + bias[co]
                                elif co < cw:
                                    y[n, oy, ox, co] = 0.0
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
    print('ok')
