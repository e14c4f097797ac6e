#!/usr/bin/env python
"""CPU walk-through of csrc/fpn.hip's index arithmetic (weight packing, coarse-window origin and tap offsets, the halo tile, the
two-output-rows-in-N MFMA form) in numpy, checked against oracle/ref_fpn.py.  A development aid for a container without a GPU:
it follows the kernel's formulas line by line (same names), so an indexing mistake shows up here before a GPU minute is spent.
Not part of the product, not a test of the HIP code itself (tests/test_hip_fpn.py is).

    python tools/sim_fpn.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_fpn  # noqa: E402

FC, TH, TW = 64, 4, 32
HR, HC = TH + 2, TW + 2
NPIX = HR * HC
CCH = 16
CS = 208
SH, SW = 6, 20
SS = SH * SW
f32 = np.float32


def np_of(nt):
    return 16 if nt == 1 else (48 if nt == 2 else 80)


def pack(w, Cout):
    T = 12 if Cout == 8 else 9
    NP = np_of(2 if Cout == 32 else 1)
    out = np.zeros((FC // 4) * T * 4 * NP, f32)
    for idx in range(out.size):
        n, kk, tap, slab = idx % NP, (idx // NP) % 4, (idx // (4 * NP)) % T, idx // (4 * NP * T)
        c = slab * 4 + kk
        if Cout == 8:
            j, kx, h, co = tap // 3, tap % 3, n >> 3, n & 7
            ky = j - h
            if n < 16 and 0 <= ky <= 2:
                out[idx] = w[co, c, ky, kx]
        elif n < Cout:
            out[idx] = w.reshape(Cout, FC, 9)[n, c, tap]
    return out, T, NP


def level(prev, lat, w_in, b_in, wp, T, NP, scale, shift, CK):
    N, _, h, w = prev.shape
    H, W = 2 * h, 2 * w
    rows2 = CK == 8
    NT = 2 if CK == 32 else 1
    WCH = 4 * T * 4 * NP
    out = np.zeros((N, H, W, CK), f32)
    intra = np.zeros((N, FC, H, W), f32)
    sy = f32(h - 1) / f32(H - 1)
    sx = f32(w - 1) / f32(W - 1)
    worst = [0, 0]
    for img in range(N):
        for by in range((H + TH - 1) // TH):
            for bx in range((W + TW - 1) // TW):
                x0, y0 = bx * TW, by * TH
                wy0 = int(sy * f32(max(y0 - 1, 0)))
                wx0 = int(sx * f32(max(x0 - 1, 0)))
                acc = np.zeros((4, 2, NT, 16, 16), f32)          # [wave][mtile][ntile][m][n]
                for cc in range(FC // CCH):
                    s_src = np.zeros(CCH * SS, f32)
                    for idx in range(CCH * SS):
                        c, r = idx // SS, idx % SS
                        py, px = wy0 + r // SW, wx0 + r % SW
                        if py < h and px < w:
                            s_src[idx] = prev[img, cc * CCH + c, py, px]
                    s_w = wp[cc * WCH:(cc + 1) * WCH]
                    s_tile = np.zeros(CCH * CS, f32)
                    for p in range(NPIX):
                        gy, gx = y0 - 1 + p // HC, x0 - 1 + p % HC
                        if not (0 <= gy < H and 0 <= gx < W):
                            continue
                        fy, fx = sy * f32(gy), sx * f32(gx)
                        iy0, ix0 = int(fy), int(fx)
                        iy1, ix1 = iy0 + (1 if iy0 < h - 1 else 0), ix0 + (1 if ix0 < w - 1 else 0)
                        ly1, lx1 = fy - f32(iy0), fx - f32(ix0)
                        ly0, lx0 = f32(1) - ly1, f32(1) - lx1
                        worst[0] = max(worst[0], iy1 - wy0)
                        worst[1] = max(worst[1], ix1 - wx0)
                        assert iy0 - wy0 >= 0 and ix0 - wx0 >= 0
                        ry0, ry1, rx0, rx1 = min(iy0 - wy0, SH - 1), min(iy1 - wy0, SH - 1), min(ix0 - wx0, SW - 1), min(ix1 - wx0, SW - 1)
                        o00, o01, o10, o11 = ry0 * SW + rx0, ry0 * SW + rx1, ry1 * SW + rx0, ry1 * SW + rx1
                        lv = lat[img, :, gy, gx]
                        for c in range(CCH):
                            ch = cc * CCH + c
                            lin = b_in[ch] + np.dot(w_in[ch], lv)
                            S = s_src[c * SS:(c + 1) * SS]
                            up = ly0 * (lx0 * S[o00] + lx1 * S[o01]) + ly1 * (lx0 * S[o10] + lx1 * S[o11])
                            s_tile[c * CS + p] = up + lin
                    for idx in range(CCH * TH * TW):
                        col, row, c = idx % TW, (idx // TW) % TH, idx // (TW * TH)
                        if y0 + row < H and x0 + col < W:
                            intra[img, cc * CCH + c, y0 + row, x0 + col] = s_tile[c * CS + (row + 1) * HC + col + 1]
                    i16 = np.arange(16)
                    for wv in range(4):
                        for ks in range(CCH // 4):
                            for kk in range(4):
                                abase = (ks * 4 + kk) * CS
                                bbase = (ks * T * 4 + kk) * NP
                                if rows2:
                                    pq, mt = wv >> 1, wv & 1
                                    for j in range(4):
                                        for kx in range(3):
                                            a = s_tile[abase + i16 + (2 * pq + j) * HC + mt * 16 + kx]
                                            b = s_w[bbase + i16 + (j * 3 + kx) * 4 * NP]
                                            acc[wv, 0, 0] += np.outer(a, b)
                                else:
                                    for ky in range(3):
                                        for kx in range(3):
                                            for t in range(2):
                                                a = s_tile[abase + i16 + (wv + ky) * HC + t * 16 + kx]
                                                for n in range(NT):
                                                    b = s_w[bbase + i16 + (ky * 3 + kx) * 4 * NP + n * 16]
                                                    acc[wv, t, n] += np.outer(a, b)
                # epilogue
                for wv in range(4):
                    for lane in range(64):
                        l16, kk = lane & 15, lane >> 4
                        for r in range(4):
                            m = 4 * kk + r
                            if rows2:
                                pq, mt, co = wv >> 1, wv & 1, l16 & 7
                                yy, xx = y0 + 2 * pq + (l16 >> 3), x0 + mt * 16 + m
                                if yy < H and xx < W:
                                    out[img, yy, xx, co] = acc[wv, 0, 0, m, l16] * scale[co] + shift[co]
                            else:
                                yy = y0 + wv
                                for n in range(NT):
                                    co = n * 16 + l16
                                    for t in range(2):
                                        xx = x0 + t * 16 + m
                                        if yy < H and xx < W:
                                            out[img, yy, xx, co] = acc[wv, t, n, m, l16] * scale[co] + shift[co]
    out = out / (1 + np.exp(-out))
    return intra, out, worst


def main():
    torch.manual_seed(3)
    sd = {}
    chs = [8, 16, 32, 64]
    g = torch.Generator().manual_seed(5)

    def rnd(*s, k=1.0):
        return torch.randn(*s, generator=g) * k

    sd["out0.0.weight"], sd["out0.0.bias"] = rnd(64, 64, 1, 1, k=0.1), rnd(64, k=0.1)
    for k, ck in ((1, 32), (2, 16), (3, 8)):
        sd["inner%d.weight" % k], sd["inner%d.bias" % k] = rnd(64, ck, 1, 1, k=0.2), rnd(64, k=0.1)
        sd["out%d.0.weight" % k], sd["out%d.0.bias" % k] = rnd(ck, 64, 3, 3, k=0.05), rnd(ck, k=0.1)
    for k, ck in ((0, 64), (1, 32), (2, 16), (3, 8)):
        sd["out%d.1.weight" % k] = 0.5 + torch.rand(ck, generator=g)
        sd["out%d.1.bias" % k] = rnd(ck, k=0.2)
        sd["out%d.1.running_mean" % k] = rnd(ck, k=0.3)
        sd["out%d.1.running_var" % k] = 0.5 + torch.rand(ck, generator=g)
    N, h, w = 1, int(os.environ.get("SIM_H", 3)), int(os.environ.get("SIM_W", 9))
    feats = ref_fpn.make_case(11, N, h, w)
    want = ref_fpn.fpn_decoder_forward(sd, *feats)
    prev = feats[3].numpy()
    for k, ck, lat in ((1, 32, feats[2]), (2, 16, feats[1]), (3, 8, feats[0])):
        wp, T, NP = pack(sd["out%d.0.weight" % k].numpy(), ck)
        bn = "out%d.1." % k
        scale = (sd[bn + "weight"] / torch.sqrt(sd[bn + "running_var"] + 1e-5)).numpy()
        shift = (sd[bn + "bias"] + (sd["out%d.0.bias" % k] - sd[bn + "running_mean"]) * torch.from_numpy(scale)).numpy()
        intra, out, worst = level(prev, lat.numpy(), sd["inner%d.weight" % k].numpy().reshape(64, ck), sd["inner%d.bias" % k].numpy(),
                                  wp, T, NP, scale, shift, ck)
        err = np.abs(out.transpose(0, 3, 1, 2) - want[k].numpy()).max()
        print("level %d (Ck=%2d, %dx%d): max abs err %.2e, window rows/cols used %d/%d of %d/%d" % (
            k, ck, intra.shape[2], intra.shape[3], err, worst[0] + 1, worst[1] + 1, SH, SW))
        assert err < 1e-4, err
        prev = intra
    print("ok")


if __name__ == "__main__":
    main()
