#!/usr/bin/env python
"""numpy walk-through of csrc/conv3d_x3.hip (the regularizer's 3x3x3 convolutions in THREE-TERM bf16 split form): what every block /
wavefront / lane computes, restated index by index, and the accuracy the form has against an fp64 convolution next to a plain fp32
accumulation.  CPU only; tests/test_kernel_walkthroughs.py runs it.  (The HIP kernel is tested by tests/test_hip_x3.py.)

    1. split3():  v == h + m + l EXACTLY for every fp32 v below bf16's overflow threshold (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m), RNE);
    2. x3_pack_kernel's layout  packed[((ct*NCH + chunk)*3 + kd)*STEPS + step][term][lane][8];
    3. the kernel's loop nest: input-plane sweep with three accumulator sets (od = p + 1 - kd), staging box with the even-columns-first
       order of the stride-2 instances, K blocks q = 4*step + (lane >> 4) -> (tap, channel octet), six MFMAs per (step, row) in the
       kernel's order, fp32 accumulation;
    4. the error of (3) against fp64, next to an fp32 direct accumulation of the same convolution.
"""
import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


def bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=f32)).to(torch.bfloat16).float().numpy()


def split3(v):
    v = np.asarray(v, f32)
    h = bf16(v)
    r = (v - h).astype(f32)
    m = bf16(r)
    r2 = (r - m).astype(f32)
    return h, m, bf16(r2)


class Cfg:
    def __init__(self, CK, SHW, NT, MTB):
        self.CK, self.SHW, self.NT, self.MTB = CK, SHW, NT, MTB
        self.KQ = CK // 8
        self.NKB = 9 * self.KQ
        self.STEPS = (self.NKB + 3) // 4
        self.TH, self.TW = 4 * NT, 16
        self.BH = SHW * self.TH + (3 - SHW)
        self.BWC = SHW * self.TW + (3 - SHW)
        self.EV = (self.BWC + 1) // 2

    def col_index(self, c):
        return c if self.SHW == 1 else (c & 1) * self.EV + (c >> 1)


def pack(w, cfg):
    """x3_pack_kernel: w [Cout][Cin][3][3][3] -> [ct][chunk][kd][step][term][lane][8] (bf16 values held as fp32)"""
    Cout, Cin = w.shape[:2]
    NCH, CT = Cin // cfg.CK, (Cout + 15) // 16
    out = np.zeros((CT, NCH, 3, cfg.STEPS, 3, 64, 8), f32)
    for ct in range(CT):
        for chunk in range(NCH):
            for kd in range(3):
                for step in range(cfg.STEPS):
                    for lane in range(64):
                        m, q = ct * 16 + (lane & 15), 4 * step + (lane >> 4)
                        if q >= cfg.NKB or m >= Cout:
                            continue
                        tap9, c0 = q // cfg.KQ, chunk * cfg.CK + (q % cfg.KQ) * 8
                        v = w[m, c0:c0 + 8, kd, tap9 // 3, tap9 % 3]
                        for t, part in enumerate(split3(v)):
                            out[ct, chunk, kd, step, t, lane] = part
    return out


def mfma(a, b, c):
    """v_mfma_f32_16x16x32_bf16: c[m][n] += sum_k a[m][k] b[n][k]; products of bf16 are exact, the sum is rounded once to fp32 here
    (the hardware's internal order is not architected; the bound used by the tests does not depend on it)."""
    return (c.astype(np.float64) + a.astype(np.float64) @ b.astype(np.float64).T).astype(f32)


def conv_x3(x, w, cfg):
    """x [Cin][D][H][W], w [Cout][Cin][3][3][3] -> [Cout][D][Ho][Wo], stride (1, SHW, SHW), padding 1 - as the kernel's blocks do it."""
    Cin, D, H, W = x.shape
    Cout = w.shape[0]
    S, NT, CK = cfg.SHW, cfg.NT, cfg.CK
    Ho, Wo = (H - 1) // S + 1, (W - 1) // S + 1
    NCH = Cin // CK
    wp = pack(w, cfg)
    y = np.zeros((Cout, D, Ho, Wo), f32)
    for ty in range((Ho + cfg.TH - 1) // cfg.TH):
        for tx in range((Wo + 15) // 16):
            y0, x0 = ty * cfg.TH, tx * 16
            for ct in range(wp.shape[0]):
                # acc[set][wave][row][pixel m][cout n]; set 0 = output plane p-1, 1 = p, 2 = p+1
                acc = np.zeros((3, 4, NT, 16, 16), f32)
                for p in range(D):
                    kd_lo, kd_hi = max(0, p + 2 - D), min(2, p + 1)
                    for chunk in range(NCH):
                        # staging: box[row][column slot][term][channel], zero outside the image
                        box = np.zeros((cfg.BH, cfg.BWC, 3, CK), f32)
                        for r in range(cfg.BH):
                            for c in range(cfg.BWC):
                                gy, gx = y0 * S - 1 + r, x0 * S - 1 + c
                                if 0 <= gy < H and 0 <= gx < W:
                                    hml = split3(x[chunk * CK:(chunk + 1) * CK, p, gy, gx])
                                    for t in range(3):
                                        box[r, cfg.col_index(c), t] = hml[t]
                        for kd in range(kd_lo, kd_hi + 1):
                            SET = 2 - kd
                            for step in range(cfg.STEPS):
                                for wave in range(4):
                                    for nt in range(NT):
                                        # A operand: activations, M = 16 pixels of row (wave*NT + nt), K = 4 K blocks x 8 channels
                                        a = np.zeros((3, 16, 32), f32)
                                        b = np.zeros((3, 16, 32), f32)
                                        for kb in range(4):
                                            q = min(4 * step + kb, cfg.NKB - 1)       # (the weights of q >= NKB are zero)
                                            tap9, oct_ = q // cfg.KQ, q % cfg.KQ
                                            kh, kw = tap9 // 3, tap9 % 3
                                            for n in range(16):
                                                a[:, n, kb * 8:kb * 8 + 8] = box[S * (wave * NT + nt) + kh, cfg.col_index(S * n + kw), :, oct_ * 8:oct_ * 8 + 8]
                                            b[:, :, kb * 8:kb * 8 + 8] = wp[ct, chunk, kd, step, :, kb * 16:kb * 16 + 16]
                                        xh, xm, xl = a
                                        wh, wm, wl = b
                                        c = acc[SET, wave, nt]
                                        for aa, bb in ((xm, wm), (xh, wl), (xl, wh), (xh, wm), (xm, wh), (xh, wh)):   # smallest products first
                                            c = mfma(aa, bb, c)
                                        acc[SET, wave, nt] = c
                    if p - 1 >= 0:
                        store(y, acc[0], p - 1, y0, x0, ct, NT)
                    acc[0], acc[1], acc[2] = acc[1].copy(), acc[2].copy(), 0
                store(y, acc[0], D - 1, y0, x0, ct, NT)
    return y


def store(y, a, od, y0, x0, ct, NT):
    Cout, _, Ho, Wo = y.shape
    for wave in range(4):
        for nt in range(NT):
            gy = y0 + wave * NT + nt
            if gy >= Ho:
                continue
            for m in range(16):
                for n in range(16):
                    if x0 + m < Wo and ct * 16 + n < Cout:
                        y[ct * 16 + n, od, gy, x0 + m] = a[wave, nt, m, n]


def main():
    rng = np.random.default_rng(0)
    # 1. exactness of the split (also across the exponent range and for values whose parts change sign)
    v = np.concatenate([rng.standard_normal(200000).astype(f32) * f32(10.0) ** rng.integers(-20, 20, 200000).astype(f32),
                        np.array([0, 1, -1, 1.00390625, 3.3e38, 1e-30, 255.99998], f32)])     # (bf16(v) overflows above 3.3895e38)
    h, m, l = split3(v)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(f32), v) and np.array_equal(h.astype(np.float64) + m + l, v.astype(np.float64))
    print("split3: h + m + l == v exactly for %d values" % v.size)
    for name, cfg, Cin, Cout, D, H, W in (("stride 1, <16,1,2,2>", Cfg(16, 1, 2, 2), 32, 32, 3, 7, 19),
                                          ("stride 2, <8,2,4,1>", Cfg(8, 2, 4, 1), 8, 16, 3, 9, 37),
                                          ("stride 2, <16,2,2,1>", Cfg(16, 2, 2, 1), 16, 16, 2, 6, 21)):
        x = rng.standard_normal((Cin, D, H, W)).astype(f32)
        w = (rng.standard_normal((Cout, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(f32)
        got = conv_x3(x, w, cfg)
        s = cfg.SHW
        ref64 = F.conv3d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), stride=(1, s, s), padding=1)[0].numpy()
        ref32 = F.conv3d(torch.from_numpy(x)[None], torch.from_numpy(w), stride=(1, s, s), padding=1)[0].numpy()
        scale = np.abs(ref64).max()
        e3, e32 = np.abs(got - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
        print("%s  Cin=%d Cout=%d %dx%dx%d: split form %.2e of scale vs fp64, fp32 accumulation %.2e" % (name, Cin, Cout, D, H, W, e3, e32))
        assert got.shape == ref64.shape and e3 < 3 * e32 + 2e-7, (e3, e32)
    print("ok")


if __name__ == "__main__":
    main()
