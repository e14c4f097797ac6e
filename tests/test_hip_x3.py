"""The regularizer's convolutions in three-term bf16 split form (csrc/conv3d_x3.hip) against fp64 ``F.conv3d``: the split form must be
as close to the exact convolution as the fp32-MFMA kernels are (it drops terms <= 2^-24 of a product), on ragged shapes, with and
without the fused BatchNorm / ReLU / residual epilogue."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


CASES = [  # cin, cout, (sd, shw), D, H, W
    (16, 16, (1, 1), 4, 5, 70), (32, 32, (1, 1), 3, 4, 12), (64, 64, (1, 1), 2, 3, 66), (16, 32, (1, 1), 1, 17, 16), (32, 16, (1, 1), 5, 16, 33),
    (64, 64, (1, 1), 6, 20, 20), (16, 16, (1, 1), 8, 48, 64),
    (8, 16, (1, 2), 3, 16, 40), (8, 16, (1, 2), 1, 9, 35), (16, 32, (1, 2), 4, 17, 30), (32, 64, (1, 2), 2, 6, 18), (32, 64, (1, 2), 5, 32, 64),
    (16, 32, (1, 2), 8, 40, 56),
]


@pytest.mark.parametrize("cin,cout,stride,D,H,W", CASES)
@pytest.mark.parametrize("epilogue", [False, True])
def test_conv3d_x3_vs_fp64(dev, cin, cout, stride, D, H, W, epilogue):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(cin * 7 + cout + D)
    x = torch.randn(2, cin, D, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / (27 * cin) ** 0.5
    scale = torch.rand(cout, generator=gen) + 0.5 if epilogue else None
    shift = torch.randn(cout, generator=gen) if epilogue else None
    s3 = (stride[0], stride[1], stride[1])
    want = F.conv3d(x.double(), w.double(), stride=s3, padding=1)
    res = torch.randn(want.shape, generator=gen) if epilogue else None
    if epilogue:
        want = torch.relu(want * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    g = lambda t: None if t is None else t.to(dev).contiguous()
    assert ops.conv3d_x3_supported(cin, cout, stride)
    pk = ops.conv3d_x3_pack(g(w), stride)
    got = ops.conv3d_x3(g(x), pk, cin, cout, stride, g(scale), g(shift), g(res), relu=epilogue)
    assert got.shape == want.shape
    err = (got.cpu().double() - want).abs().max().item() / want.abs().max().item()
    # the fp32-MFMA kernel on the same operands, as the yardstick: the split form may not be worse than 3x its error (+ 2e-7: a few ulps of the largest output)
    ref32 = ops.conv3d(g(x), ops.conv3d_pack(g(w), False), cin, cout, stride, g(scale), g(shift), g(res), relu=epilogue)
    err32 = (ref32.cpu().double() - want).abs().max().item() / want.abs().max().item()
    assert err < 3 * err32 + 2e-7, (err, err32)
    assert err < 2e-6, err


DECONV_CASES = [(64, 32, 2, 3, 6), (32, 16, 3, 4, 20), (16, 8, 2, 5, 34), (16, 8, 5, 16, 32), (32, 16, 8, 9, 18), (64, 32, 4, 16, 16), (16, 16, 1, 8, 16)]


@pytest.mark.parametrize("cin,cout,D,H,W", DECONV_CASES)
@pytest.mark.parametrize("epilogue", [False, True])
def test_deconv3d_x3_vs_fp64(dev, cin, cout, D, H, W, epilogue):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(cin * 5 + cout + D)
    x = torch.randn(2, cin, D, H, W, generator=gen)
    w = torch.randn(cin, cout, 3, 3, 3, generator=gen) / (7 * cin) ** 0.5
    scale = torch.rand(cout, generator=gen) + 0.5 if epilogue else None
    shift = torch.randn(cout, generator=gen) if epilogue else None
    want = F.conv_transpose3d(x.double(), w.double(), stride=(1, 2, 2), padding=1, output_padding=(0, 1, 1))
    res = torch.randn(want.shape, generator=gen) if epilogue else None
    if epilogue:
        want = torch.relu(want * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    g = lambda t: None if t is None else t.to(dev).contiguous()
    assert ops.deconv3d_x3_supported(cin, cout, 1)
    pk = ops.deconv3d_x3_pack(g(w), 1)
    got = ops.deconv3d_x3(g(x), pk, cin, cout, 1, g(scale), g(shift), g(res), relu=epilogue)
    assert got.shape == want.shape
    err = (got.cpu().double() - want).abs().max().item() / want.abs().max().item()
    ref32 = ops.deconv3d(g(x), ops.conv3d_pack(g(w), True, 1), cin, cout, 1, g(scale), g(shift), g(res), relu=epilogue)
    err32 = (ref32.cpu().double() - want).abs().max().item() / want.abs().max().item()
    assert err < 3 * err32 + 2e-7, (err, err32)
    assert err < 2e-6, err


# ---------------------------------------------------------------------------------------------------------------------------------
# Depth segments: small H x W make the launchers cut D into segments; non-power-of-two depths give uneven splits and - before
# the fix - EMPTY segments whose blocks rewrote plane D-1.  Every case is checked against fp64 and for run-to-run equality.
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [9, 12, 17, 20, 40])
@pytest.mark.parametrize("cin,cout,stride,H,W", [(16, 16, (1, 1), 16, 16), (8, 16, (1, 2), 20, 24), (32, 64, (1, 2), 12, 40), (64, 64, (1, 1), 8, 20)])
def test_conv3d_x3_depth_segments(dev, D, cin, cout, stride, H, W):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(D * 131 + cin)
    x = torch.randn(1, cin, D, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / (27 * cin) ** 0.5
    shift = torch.randn(cout, generator=gen) + 3.0              # relu(shift) != 0: a stray all-zero-accumulator store would be visible
    scale = torch.rand(cout, generator=gen) + 0.5
    want = torch.relu(F.conv3d(x.double(), w.double(), stride=(stride[0], stride[1], stride[1]), padding=1) * scale.double().view(1, -1, 1, 1, 1)
                      + shift.double().view(1, -1, 1, 1, 1))
    pk = ops.conv3d_x3_pack(w.to(dev), stride)
    xs, sc, sh = x.to(dev), scale.to(dev), shift.to(dev)
    got = ops.conv3d_x3(xs, pk, cin, cout, stride, sc, sh, None, relu=True)
    assert (got.cpu().double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    for _ in range(5):
        assert torch.equal(got, ops.conv3d_x3(xs, pk, cin, cout, stride, sc, sh, None, relu=True))


@pytest.mark.parametrize("D", [9, 12, 17, 20, 40])
@pytest.mark.parametrize("cin,cout,H,W", [(32, 16, 8, 16), (64, 32, 6, 10), (16, 8, 9, 12)])
def test_deconv3d_x3_depth_segments(dev, D, cin, cout, H, W):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(D * 17 + cin)
    x = torch.randn(1, cin, D, H, W, generator=gen)
    w = torch.randn(cin, cout, 3, 3, 3, generator=gen) / (7 * cin) ** 0.5
    shift = torch.randn(cout, generator=gen) + 3.0
    scale = torch.rand(cout, generator=gen) + 0.5
    want = torch.relu(F.conv_transpose3d(x.double(), w.double(), stride=(1, 2, 2), padding=1, output_padding=(0, 1, 1))
                      * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1))
    pk = ops.deconv3d_x3_pack(w.to(dev), 1)
    xs, sc, sh = x.to(dev), scale.to(dev), shift.to(dev)
    got = ops.deconv3d_x3(xs, pk, cin, cout, 1, sc, sh, None, relu=True)
    assert (got.cpu().double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    for _ in range(5):
        assert torch.equal(got, ops.deconv3d_x3(xs, pk, cin, cout, 1, sc, sh, None, relu=True))


# ---------------------------------------------------------------------------------------------------------------------------------
# The edges of the split form's arithmetic.  Contract (include/mvs_hip.h, "Arithmetic"):
#   * finite fp32 inputs of any magnitude whose exact products and sums stay inside the fp32 range give the fp32-equivalent result:
#     h + m + l == v exactly for every normal v, including |v| beyond the largest finite bf16 (h is clamped to it; m carries the rest);
#   * subnormal operands may be flushed to zero by the matrix cores: absolute error <= K * 2^-126 * max|other operand|;
#   * a non-finite input makes exactly the outputs whose receptive field contains it non-finite (NaN where fp32 arithmetic gives +-Inf).
# ---------------------------------------------------------------------------------------------------------------------------------
def _x3_vs_fp64(dev, x, w, stride=(1, 1)):
    from mvsformer_amd import ops
    cout, cin = w.shape[:2]
    want = F.conv3d(x.double(), w.double(), stride=(stride[0], stride[1], stride[1]), padding=1)
    got = ops.conv3d_x3(x.to(dev), ops.conv3d_x3_pack(w.to(dev), stride), cin, cout, stride, None, None, None, relu=False).cpu().double()
    ref32 = ops.conv3d(x.to(dev), ops.conv3d_pack(w.to(dev), False), cin, cout, stride, None, None, None, relu=False).cpu().double()
    return got, ref32, want


@pytest.mark.parametrize("exp_x,exp_w", [(100, 0), (-100, 0), (60, 60), (-60, -60), (100, -100), (0, -100)])
def test_x3_scaled_operands(dev, exp_x, exp_w):
    """Inputs / weights scaled by 2^+-100: the split is exponent-invariant while every term stays normal."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 3, 20, 36, generator=gen) * 2.0 ** exp_x
    w = torch.randn(32, 16, 3, 3, 3, generator=gen) / (27 * 16) ** 0.5 * 2.0 ** exp_w
    got, ref32, want = _x3_vs_fp64(dev, x, w)
    s = want.abs().max().item()
    assert s > 0 and torch.isfinite(got).all()
    err, err32 = (got - want).abs().max().item() / s, (ref32 - want).abs().max().item() / s
    assert err < 3 * err32 + 2e-7 and err < 2e-6, (err, err32)


def test_x3_subnormal_inputs(dev):
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(1, 16, 2, 16, 32, generator=gen) * 2.0 ** -130            # every value subnormal (|v| < 2^-126) or zero
    w = torch.randn(16, 16, 3, 3, 3, generator=gen)
    got, _, want = _x3_vs_fp64(dev, x, w)
    bound = 27 * 16 * 2.0 ** -126 * w.abs().max().item()
    assert torch.isfinite(got).all() and (got - want).abs().max().item() <= bound
    # normal inputs with a few subnormals mixed in: the subnormals' contribution is below the bound, everything else is exact as usual
    x2 = torch.randn(1, 16, 2, 16, 32, generator=gen)
    x2[0, :, :, ::3, ::5] = x[0, :, :, ::3, ::5]
    got2, ref32, want2 = _x3_vs_fp64(dev, x2, w)
    s = want2.abs().max().item()
    assert (got2 - want2).abs().max().item() / s < 3 * (ref32 - want2).abs().max().item() / s + 2e-7


def test_x3_near_flt_max(dev):
    """|v| above the largest finite bf16 (3.3895e38): bf16(v) would round to +-Inf; the staging clamps h to the largest finite bf16 and m, l
    carry the remainder exactly, so the result is the fp32-equivalent one (the exact sums here stay far inside the fp32 range)."""
    gen = torch.Generator().manual_seed(7)
    fmax = torch.finfo(torch.float32).max
    x = torch.randn(1, 16, 2, 16, 32, generator=gen)
    x[0, 3, 1, 5, 7] = fmax
    x[0, 5, 0, 9, 20] = -fmax
    x[0, 0, 1, 2, 30] = 3.39e38                                               # between bf16 max and FLT_MAX
    x[0, 9, 0, 12, 3] = float(torch.tensor(3.3895e38).to(torch.bfloat16).float())     # exactly the largest finite bf16
    w = torch.randn(16, 16, 3, 3, 3, generator=gen) * 1e-3                    # |x*w| <= 3.4e35 * few
    got, ref32, want = _x3_vs_fp64(dev, x, w)
    assert torch.isfinite(got).all()
    s = want.abs().max().item()
    err, err32 = (got - want).abs().max().item() / s, (ref32 - want).abs().max().item() / s
    assert err < 3 * err32 + 2e-7 and err < 2e-6, (err, err32)


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_x3_nonfinite_propagation(dev, bad):
    """A non-finite input voxel poisons exactly its receptive field (3 x 3 x 3 output voxels x all output channels, weights all non-zero),
    as in the fp32 kernel; the value is NaN or +-Inf (fp32: inf * w = +-inf; split form: inf - inf in the remainder terms gives NaN)."""
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(1, 16, 4, 20, 36, generator=gen)
    x[0, 5, 2, 10, 17] = bad
    w = torch.randn(16, 16, 3, 3, 3, generator=gen) * 0.1 + 0.5 * torch.sign(torch.randn(16, 16, 3, 3, 3, generator=gen))
    got, ref32, _ = _x3_vs_fp64(dev, x, w)
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref32))
    mask = torch.zeros(4, 20, 36, dtype=torch.bool)
    mask[1:4, 9:12, 16:19] = True
    assert torch.equal(~torch.isfinite(got[0]), mask.unsqueeze(0).expand(16, -1, -1, -1))


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 32)])
def test_x3_all_positive_worst_case_sums(dev, cin, cout):
    """No cancellation: 27 * Cin positive products per output (1728 at Cin = 64), the case where accumulated rounding is largest relative
    to nothing - the split form must stay within 3x the fp32-MFMA kernel's own error against fp64."""
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(1, cin, 3, 12, 40, generator=gen) + 0.5
    w = (torch.rand(cout, cin, 3, 3, 3, generator=gen) + 0.5) / (27 * cin)
    got, ref32, want = _x3_vs_fp64(dev, x, w)
    s = want.abs().max().item()
    err, err32 = (got - want).abs().max().item() / s, (ref32 - want).abs().max().item() / s
    assert err < 3 * err32 + 2e-7 and err < 2e-6, (err, err32)


# ---------------------------------------------------------------------------------------------------------------------------------
# CostRegNet3D's tail (conv11 + BatchNorm + ReLU + skip + 1x1x1 prob) in split form, csrc/tail_x3.hip (models/module.py:575-592)
# ---------------------------------------------------------------------------------------------------------------------------------
TAIL_CASES = [(1, 1, 1, 1), (1, 2, 3, 5), (2, 4, 8, 16), (1, 3, 9, 17), (2, 5, 7, 33), (1, 4, 24, 40), (1, 8, 16, 48), (1, 9, 8, 16), (1, 17, 5, 20),
              (1, 20, 12, 12), (1, 40, 4, 16)]


@pytest.mark.parametrize("B,D,H,W", TAIL_CASES)
@pytest.mark.parametrize("full", [True, False])
def test_tail_x3_vs_fp64(dev, B, D, H, W, full):
    """``full``: folded BatchNorm + skip tensor + bias (what CostRegNet3D.logits passes); else every optional pointer NULL."""
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + D * 100 + H + W)
    x = torch.randn(B, 16, D, H, W, generator=gen)
    w = torch.randn(16, 8, 3, 3, 3, generator=gen) / (7 * 16) ** 0.5
    scale = torch.rand(8, generator=gen) + 0.5 if full else None
    shift = torch.randn(8, generator=gen) if full else None
    res = torch.randn(B, 8, D, 2 * H, 2 * W, generator=gen) if full else None
    pw = torch.randn(8, generator=gen)
    pb = torch.randn(1, generator=gen) if full else None
    y = F.conv_transpose3d(x.double(), w.double(), stride=(1, 2, 2), padding=1, output_padding=(0, 1, 1))
    if full:
        y = y * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    y = torch.relu(y)
    if full:
        y = y + res.double()
    want = (y * pw.double().view(1, 8, 1, 1, 1)).sum(1) + (pb.double() if full else 0.0)
    g = lambda t: None if t is None else t.to(dev).contiguous()
    got = ops.tail_x3(g(x), ops.tail_x3_pack(g(w)), g(scale), g(shift), g(res), g(pw), g(pb), relu=True)
    assert got.shape == (B, D, 2 * H, 2 * W)
    s = want.abs().max().item()
    err = (got.cpu().double() - want).abs().max().item() / s
    ref32 = ops.deconv3d_prob1(g(x), ops.conv3d_pack(g(w), True, 1), 16, g(scale), g(shift), g(res), g(pw), g(pb), relu=True) if W % 4 == 0 else None
    if ref32 is not None:
        err32 = (ref32.cpu().double() - want).abs().max().item() / s
        assert err < 3 * err32 + 2e-7, (err, err32)
    assert err < 2e-6, err
    for _ in range(3):
        assert torch.equal(got, ops.tail_x3(g(x), ops.tail_x3_pack(g(w)), g(scale), g(shift), g(res), g(pw), g(pb), relu=True))


def test_costregnet3d_logits_paths_agree(dev):
    """CostRegNet3D.logits through the split-form tail, the fp32-MFMA fused tail and the unfused conv11 + prob: same logits to fp32 rounding."""
    import os
    import mvsformer_amd as m
    torch.manual_seed(3)
    net = m.CostRegNet3D(8, 8).eval()
    m.randomize_bn_(net, 5)
    net = net.to(dev)
    x = torch.randn(1, 8, 4, 64, 96, device=dev)
    outs = []
    for env in ({"MVS_CONV_X3_MIN_VOXELS": "0"}, {"MVS_TAIL": "fp32"}, {"MVS_FUSE_PROB": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            net._dcache = {}
            outs.append(net.logits(x))
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    s = outs[2].abs().max().item()
    assert (outs[0] - outs[2]).abs().max().item() < 5e-6 * s
    assert (outs[1] - outs[2]).abs().max().item() < 5e-6 * s


# ---------------------------------------------------------------------------------------------------------------------------------
# Small-volume split form (csrc/conv3d_x3_small.hip): CostRegNet's inner layers, Conv3d stride (1,1,1) / (2,2,2) and the stride-2
# ConvTranspose3d (models/module.py:469-505), against fp64 and the fp32-MFMA kernels
# ---------------------------------------------------------------------------------------------------------------------------------
SMALL_CONV = [(8, 16, 2, 8, 16, 24), (16, 32, 2, 5, 9, 35), (32, 64, 2, 4, 18, 24), (32, 32, 1, 8, 12, 20), (64, 64, 1, 4, 18, 24),
              (64, 64, 1, 2, 3, 5), (16, 16, 1, 3, 7, 33), (8, 8, 1, 1, 1, 1), (16, 8, 2, 6, 6, 6)]


@pytest.mark.parametrize("cin,cout,stride,D,H,W", SMALL_CONV)
@pytest.mark.parametrize("epilogue", [False, True])
def test_conv3d_small_vs_fp64(dev, cin, cout, stride, D, H, W, epilogue):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(cin * 3 + cout + D + W)
    x = torch.randn(2, cin, D, H, W, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / (27 * cin) ** 0.5
    scale = torch.rand(cout, generator=gen) + 0.5 if epilogue else None
    shift = torch.randn(cout, generator=gen) if epilogue else None
    want = F.conv3d(x.double(), w.double(), stride=stride, padding=1)
    res = torch.randn(want.shape, generator=gen) if epilogue else None
    if epilogue:
        want = torch.relu(want * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    g = lambda t: None if t is None else t.to(dev).contiguous()
    assert ops.conv3d_small_supported(cin, cout, stride, False)
    pk = ops.conv3d_small_pack(g(w), stride, False)
    got = ops.conv3d_small(g(x), pk, cin, cout, stride, False, g(scale), g(shift), g(res), relu=epilogue)
    assert got.shape == want.shape
    s = want.abs().max().item()
    err = (got.cpu().double() - want).abs().max().item() / s
    ref32 = ops.conv3d(g(x), ops.conv3d_pack(g(w), False), cin, cout, (stride, stride), g(scale), g(shift), g(res), relu=epilogue)
    err32 = (ref32.cpu().double() - want).abs().max().item() / s
    assert err < 3 * err32 + 2e-7, (err, err32)
    assert err < 2e-6, err
    assert torch.equal(got, ops.conv3d_small(g(x), pk, cin, cout, stride, False, g(scale), g(shift), g(res), relu=epilogue))


SMALL_DECONV = [(64, 32, 4, 18, 24), (32, 16, 8, 9, 12), (16, 8, 3, 5, 7), (64, 32, 1, 1, 1), (32, 16, 2, 4, 17), (16, 16, 5, 6, 34)]


@pytest.mark.parametrize("cin,cout,D,H,W", SMALL_DECONV)
@pytest.mark.parametrize("epilogue", [False, True])
def test_deconv3d_small_vs_fp64(dev, cin, cout, D, H, W, epilogue):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(cin * 5 + cout + D + W)
    x = torch.randn(2, cin, D, H, W, generator=gen)
    w = torch.randn(cin, cout, 3, 3, 3, generator=gen) / (4 * cin) ** 0.5
    scale = torch.rand(cout, generator=gen) + 0.5 if epilogue else None
    shift = torch.randn(cout, generator=gen) if epilogue else None
    want = F.conv_transpose3d(x.double(), w.double(), stride=2, padding=1, output_padding=1)
    res = torch.randn(want.shape, generator=gen) if epilogue else None
    if epilogue:
        want = torch.relu(want * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)) + res.double()
    g = lambda t: None if t is None else t.to(dev).contiguous()
    assert ops.conv3d_small_supported(cin, cout, 2, True)
    pk = ops.conv3d_small_pack(g(w), 2, True)
    got = ops.conv3d_small(g(x), pk, cin, cout, 2, True, g(scale), g(shift), g(res), relu=epilogue)
    assert got.shape == want.shape
    s = want.abs().max().item()
    err = (got.cpu().double() - want).abs().max().item() / s
    ref32 = ops.deconv3d(g(x), ops.conv3d_pack(g(w), True, 2), cin, cout, 2, g(scale), g(shift), g(res), relu=epilogue)
    err32 = (ref32.cpu().double() - want).abs().max().item() / s
    assert err < 3 * err32 + 2e-7, (err, err32)
    assert err < 2e-6, err
    assert torch.equal(got, ops.conv3d_small(g(x), pk, cin, cout, 2, True, g(scale), g(shift), g(res), relu=epilogue))
