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
