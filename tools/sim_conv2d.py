#!/usr/bin/env python
"""CPU walk-through of csrc/conv2d.hip's index arithmetic (weight packing incl. the row-pair form, halo staging, stride-2 fragment
addresses, epilogue lane mapping) in numpy against torch's conv2d, for the FPN encoder's eight layer shapes.  A development aid for a
container without a GPU (same idea as tools/sim_fpn.py); not part of the product and not a test of the HIP code itself.

    python tools/sim_conv2d.py
"""
import numpy as np
import torch
import torch.nn.functional as F

TH, TW = 4, 32
f32 = np.float32


def np_of(nt):
    return 16 if nt == 1 else (48 if nt == 2 else 80)


def pad_cs(raw, s):
    return raw + ((16 - raw % 32) + 32) % 32 if s == 1 else raw + (1 if raw % 2 == 0 else 0)


def pack(w, Cin, Cout, KS):
    rows2 = Cout == 8
    T = (KS + 1) * KS if rows2 else KS * KS
    NP = np_of(1 if rows2 else (Cout + 15) // 16)
    CP = (Cin + 3) // 4 * 4
    out = np.zeros((CP // 4) * T * 4 * NP, f32)
    for idx in range(out.size):
        n, kk, tap, slab = idx % NP, (idx // NP) % 4, (idx // (4 * NP)) % T, idx // (4 * NP * T)
        c = slab * 4 + kk
        if c >= Cin:
            continue
        if rows2:
            j, kx, h, co = tap // KS, tap % KS, n >> 3, n & 7
            ky = j - h
            if n < 16 and 0 <= ky < KS:
                out[idx] = w[co, c, ky, kx]
        elif n < Cout:
            out[idx] = w[n, c, tap // KS, tap % KS]
    return out, T, NP, CP


def conv(x, wp, T, NP, CP, Cin, Cout, KS, S):
    N, _, H, W = x.shape
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    rows2 = Cout == 8
    NT = 1 if rows2 else (Cout + 15) // 16
    CC = CP if CP < 8 else 8
    IR, IC = (TH - 1) * S + KS, (TW - 1) * S + KS
    CS = pad_cs(IR * IC, S)
    WCH = (CC // 4) * T * 4 * NP
    P = KS // 2
    y = np.zeros((N, Cout, Ho, Wo), f32)
    i16 = np.arange(16)
    for img in range(N):
        for by in range((Ho + TH - 1) // TH):
            for bx in range((Wo + TW - 1) // TW):
                x0, y0 = bx * TW, by * TH
                iy0, ix0 = y0 * S - P, x0 * S - P
                acc = np.zeros((4, 2, NT, 16, 16), f32)
                for ch in range(CP // CC):
                    s_in = np.zeros(CC * CS, f32)
                    for idx in range(CC * IR * IC):
                        c, r = idx // (IR * IC), idx % (IR * IC)
                        gy, gx, cin = iy0 + r // IC, ix0 + r % IC, ch * CC + c
                        if cin < Cin and 0 <= gy < H and 0 <= gx < W:
                            s_in[c * CS + r] = x[img, cin, gy, gx]
                    s_w = wp[ch * WCH:(ch + 1) * WCH]
                    for wv in range(4):
                        for ks in range(CC // 4):
                            for kk in range(4):
                                abase = (ks * 4 + kk) * CS + S * i16
                                bbase = (ks * T * 4 + kk) * NP + i16
                                if rows2:
                                    pq, mt = wv >> 1, wv & 1
                                    for j in range(KS + 1):
                                        for kx in range(KS):
                                            acc[wv, 0, 0] += np.outer(s_in[abase + (2 * pq + j) * IC + mt * 16 + kx], s_w[bbase + (j * KS + kx) * 4 * NP])
                                else:
                                    for ky in range(KS):
                                        for kx in range(KS):
                                            for t in range(2):
                                                a = s_in[abase + (wv * S + ky) * IC + t * 16 * S + kx]
                                                for n in range(NT):
                                                    acc[wv, t, n] += np.outer(a, s_w[bbase + (ky * KS + kx) * 4 * NP + n * 16])
                for wv in range(4):
                    for lane in range(64):
                        l16, kk = lane & 15, lane >> 4
                        for r in range(4):
                            m = 4 * kk + r
                            if rows2:
                                pq, mt, co = wv >> 1, wv & 1, l16 & 7
                                yy, xx = y0 + 2 * pq + (l16 >> 3), x0 + mt * 16 + m
                                if yy < Ho and xx < Wo:
                                    y[img, co, yy, xx] = acc[wv, 0, 0, m, l16]
                            else:
                                for n in range(NT):
                                    co = n * 16 + l16
                                    for t in range(2):
                                        yy, xx = y0 + wv, x0 + t * 16 + m
                                        if co < Cout and yy < Ho and xx < Wo:
                                            y[img, co, yy, xx] = acc[wv, t, n, m, l16]
    return y


def main():
    g = torch.Generator().manual_seed(0)
    for Cin, Cout, KS, S in ((3, 8, 7, 1), (8, 8, 5, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1), (32, 64, 3, 2), (64, 64, 3, 1)):
        H, W = (7, 37) if S == 1 else (9, 71)
        x = torch.randn(1, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, KS, KS, generator=g) / (Cin * KS * KS) ** 0.5
        wp, T, NP, CP = pack(w.numpy(), Cin, Cout, KS)
        got = conv(x.numpy(), wp, T, NP, CP, Cin, Cout, KS, S)
        want = F.conv2d(x, w, stride=S, padding=KS // 2).numpy()
        err = np.abs(got - want).max()
        print("(%2d,%2d,%d,%d) %dx%d -> %dx%d  max abs err %.2e" % (Cin, Cout, KS, S, H, W, got.shape[2], got.shape[3], err))
        assert got.shape == want.shape and err < 1e-5
    print("ok")


if __name__ == "__main__":
    main()
