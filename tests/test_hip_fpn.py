"""GPU parity tests of the FPN decoder (csrc/fpn.hip through the C ABI; reference models/module.py:242-270).

Tolerance: the decoder is fp32 end to end; only the summation order differs from the reference's convolutions (MFMA k-blocks of
4 channels x 9 taps here), so every feature map must agree within 2e-5 of its own scale against (a) the golden vectors made by
the real module and (b) the CPU oracle on other shapes; the full-size check runs the oracle's torch ops on the GPU (fp32) as the
plain PyTorch reference of the same op and allows 1e-4 of scale.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t

TOL = 2e-5


def scale_err(got, want):
    want = torch.as_tensor(want, dtype=torch.float64)
    return float((torch.as_tensor(got, dtype=torch.float64) - want).abs().max() / want.abs().max().clamp_min(1e-6))


def build_decoder(sd=None, seed=0):
    from mvsformer_amd import FPNDecoder
    from oracle import ref_fpn
    torch.manual_seed(seed)
    dec = FPNDecoder([8, 16, 32, 64])
    if sd is None:
        ref_fpn.randomize_bn(dec, seed + 1)
    else:
        missing, unexpected = dec.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return dec.eval()


def test_fpn_decoder_has_no_cpu_path_and_is_checkpoint_compatible():
    """CPU: the parameter names are the reference's (golden state_dict loads); a forward on CPU tensors fails loudly (training mode runs
    the HIP autograd functions: there is no CPU fallback in either mode)."""
    g = load_golden("fpn_decoder.npz")
    dec = build_decoder({k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")})
    assert sorted(k for k in dec.state_dict() if not k.endswith("num_batches_tracked")) == sorted(k[3:] for k in g if k.startswith("sd."))
    from mvsformer_amd._lib import MvsHipError
    with pytest.raises(MvsHipError):
        dec.train()(*[t(g[k]) for k in ("conv01", "conv11", "conv21", "conv31")])


@pytest.mark.gpu
@pytest.mark.parametrize("x3", ["1", "strip", "0"])
def test_fpn_decoder_vs_golden(monkeypatch, x3):
    """All three forms of the full-resolution level: contraction before the upsampling (csrc/fpn_cp.hip, default), the split-form strip kernel
    (csrc/fpn_x3.hip) and the fp32-MFMA kernel (csrc/fpn.hip)."""
    monkeypatch.setenv("MVS_FPN_X3", x3)
    g = load_golden("fpn_decoder.npz")
    dev = torch.device("cuda:0")
    dec = build_decoder({k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")}).to(dev)
    outs = dec(*[t(g[k], dev) for k in ("conv01", "conv11", "conv21", "conv31")])
    for i, o in enumerate(outs):
        assert tuple(o.shape) == g["out%d" % i].shape
        assert o.permute(0, 2, 3, 1).is_contiguous()            # channel-last memory, reference's logical shape
        assert scale_err(o.cpu(), g["out%d" % i]) < TOL, i


@pytest.mark.gpu
@pytest.mark.parametrize("N,h,w", [(3, 7, 9), (1, 1, 1), (2, 2, 17), (1, 25, 2)])
@pytest.mark.parametrize("x3", ["1", "strip"])
def test_fpn_decoder_vs_oracle(N, h, w, x3, monkeypatch):
    """Partial tiles in both directions, several images, the degenerate 1x1 coarsest level (upsampling scale 0), a tall narrow image (the
    full-resolution level's 16-column strip cut into vertical segments: 200 rows)."""
    from oracle import ref_fpn
    monkeypatch.setenv("MVS_FPN_X3", x3)
    dec = build_decoder(seed=3)
    feats = ref_fpn.make_case(4, N, h, w)
    want = ref_fpn.fpn_decoder_forward({k: v.detach() for k, v in dec.state_dict().items()}, *feats)
    dec = dec.to("cuda:0")
    outs = dec(*[f.to("cuda:0") for f in feats])
    for i, (o, ww) in enumerate(zip(outs, want)):
        assert o.shape == ww.shape
        assert scale_err(o.cpu(), ww) < TOL, i


@pytest.mark.gpu
def test_fpn_features_feed_the_sweeps_without_a_copy():
    """The reference's hand-over ``feat.reshape(B,V,C,H,W)`` (mvsformer_model.py:232-235) of our channel-last maps is a view and
    ops.to_channels_last passes it through: no nchw_to_nhwc launch between decoder and sweeps."""
    from mvsformer_amd import ops
    from oracle import ref_fpn
    dec = build_decoder(seed=5).to("cuda:0")
    B, V = 2, 3
    outs = dec(*[f.to("cuda:0") for f in ref_fpn.make_case(6, B * V, 2, 3)])
    for o in outs:
        f5 = o.reshape(B, V, o.shape[1], o.shape[2], o.shape[3])
        assert f5.data_ptr() == o.data_ptr()
        cl = ops.to_channels_last(f5)
        assert cl.data_ptr() == o.data_ptr() and cl.is_contiguous() and cl.shape == (B, V, o.shape[2], o.shape[3], o.shape[1])


@pytest.mark.gpu
def test_fpn_decoder_full_size_vs_torch_on_gpu():
    """BASELINE configs[1] geometry (5 views, 1152x1536): against the oracle's torch ops run on the GPU in fp32."""
    from oracle import ref_fpn
    dev = torch.device("cuda:0")
    dec = build_decoder(seed=7).to(dev)
    feats = [f.to(dev) for f in ref_fpn.make_case(8, 5, 144, 192)]
    outs = dec(*feats)
    with torch.no_grad():
        want = ref_fpn.fpn_decoder_forward({k: v.detach() for k, v in dec.state_dict().items()}, *feats)
    for i, (o, ww) in enumerate(zip(outs, want)):
        assert o.shape == ww.shape
        err = float((o - ww).abs().max() / ww.abs().max())
        assert err < 1e-4, (i, err)
    assert np.isfinite(float(outs[3].sum()))


@pytest.mark.gpu
def test_decoder_to_cascade_hand_over_is_bit_identical():
    """Decoder -> reference-style reshape -> 4-stage cascade: the channel-last maps give exactly the depth that NCHW-contiguous
    copies of the same values give, with no transpose kernel launched."""
    import mvsformer_amd as m
    from mvsformer_amd import ops, synth
    from oracle import ref_fpn
    dev = torch.device("cuda:0")
    B, V, H, W = 1, 3, 64, 128                        # stage 1 (1/8 res) needs H, W divisible by 8 for CostRegNet
    _, proj, dv, _ = synth.make_inputs(V, H, W, seed=2, device=dev)
    dec = build_decoder(seed=9).to(dev)
    outs = dec(*[f.to(dev) for f in ref_fpn.make_case(10, B * V, H // 8, W // 8)])
    feats = {"stage%d" % (i + 1): o.reshape(B, V, o.shape[1], o.shape[2], o.shape[3]) for i, o in enumerate(outs)}
    net = m.CascadeMVS().to(dev).eval()
    m.randomize_bn_(net, seed=1)
    with ops.kernel_timer() as kt:
        a = net(feats, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    assert not any("nchw_to_nhwc" in k for k in kt.events)
    b = net({k: v.contiguous() for k, v in feats.items()}, proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])
    assert torch.equal(a["refined_depth"], b["refined_depth"]) and torch.isfinite(a["refined_depth"]).all()


# ------------------------------------------------------------------------------------------------ FPNEncoder (csrc/conv2d.hip)
def build_encoder(sd=None, seed=0):
    from mvsformer_amd import FPNEncoder
    from oracle import ref_fpn
    torch.manual_seed(seed)
    enc = FPNEncoder([8, 16, 32, 64])
    if sd is None:
        ref_fpn.randomize_bn(enc, seed + 1)
    else:
        missing, unexpected = enc.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return enc.eval()


def test_fpn_encoder_has_no_cpu_path_and_is_checkpoint_compatible():
    g = load_golden("fpn_encoder.npz")
    enc = build_encoder({k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")})
    assert sorted(k for k in enc.state_dict() if not k.endswith("num_batches_tracked")) == sorted(k[3:] for k in g if k.startswith("sd."))
    from mvsformer_amd._lib import MvsHipError
    with pytest.raises(MvsHipError):
        enc.train()(t(g["x"]))


@pytest.mark.gpu
@pytest.mark.parametrize("x3", ["1", "0"])
def test_fpn_encoder_vs_golden(monkeypatch, x3):
    """conv00 / conv01 in split form (csrc/conv2d_x3.hip, default) and on the fp32 matrix cores (csrc/conv2d.hip)."""
    monkeypatch.setenv("MVS_FPN_X3", x3)
    g = load_golden("fpn_encoder.npz")
    dev = torch.device("cuda:0")
    enc = build_encoder({k[3:]: t(v) for k, v in g.items() if k.startswith("sd.")}).to(dev)
    outs = enc(t(g["x"], dev))
    for i, o in enumerate(outs):
        assert tuple(o.shape) == g["out%d" % i].shape and o.is_contiguous()
        assert scale_err(o.cpu(), g["out%d" % i]) < TOL, i


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(2, 37, 51), (1, 8, 8), (3, 64, 200)])
def test_fpn_encoder_vs_oracle(N, H, W):
    """Odd sizes (stride-2 layers round up, partial tiles, scalar stores), a tile-less 8x8 image, several images and tiles."""
    from oracle import ref_fpn
    enc = build_encoder(seed=21)
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(22))
    want = ref_fpn.fpn_encoder_forward({k: v.detach() for k, v in enc.state_dict().items()}, x)
    outs = enc.to("cuda:0")(x.to("cuda:0"))
    for i, (o, ww) in enumerate(zip(outs, want)):
        assert o.shape == ww.shape
        assert scale_err(o.cpu(), ww) < TOL, i


@pytest.mark.gpu
def test_full_resolution_layers_run_in_split_form_by_default(monkeypatch):
    """conv00, conv01 and the decoder's last level launch the split-form kernels (csrc/conv2d_x3.hip, csrc/fpn_cp.hip) unless MVS_FPN_X3 says
    otherwise, and the decoder takes both of its channel-last sources from their producers (no torch copy kernel in between)."""
    from mvsformer_amd import ops
    monkeypatch.delenv("MVS_FPN_X3", raising=False)
    dev = torch.device("cuda:0")
    enc, dec = build_encoder(seed=41).to(dev), build_decoder(seed=42).to(dev)
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(43)).to(dev)
    with ops.kernel_timer() as kt:
        dec(*enc(x))
    names = set(kt.events)
    assert {"enc_x3_kernel<3,8,7>", "enc_x3_kernel<8,8,5>", "fpn8_cp_kernel"} <= names, names
    assert not any(k.startswith("fpn_level_kernel<8>") or k in ("conv2d_kernel<3,8,7,1>", "conv2d_kernel<8,8,5,1>", "fpn8_x3_kernel") for k in names), names
    feats = enc(x)
    assert feats[0]._mvs_nhwc[0].shape == (1, 64, 96, 8)     # conv01's channel-last companion rides on the tensor ...
    assert torch.equal(feats[0]._mvs_nhwc[0].permute(0, 3, 1, 2), feats[0])
    from mvsformer_amd.fpn import _channels_last
    assert _channels_last(feats[0]).data_ptr() == feats[0]._mvs_nhwc[0].data_ptr()
    lean = enc(x, conv01_channels_last=True)                 # what DINOMVSNet asks for: conv01 as a view of its channel-last buffer, no NCHW copy
    assert not lean[0].is_contiguous() and lean[0].permute(0, 2, 3, 1).is_contiguous()
    assert all(torch.equal(a, b) for a, b in zip(lean, feats))
    assert all(torch.equal(a, b) for a, b in zip(dec(*lean), dec(*feats)))
    feats[0].mul_(2.0)                                       # ... and is dropped once the tensor was written to
    assert _channels_last(feats[0]).data_ptr() != feats[0]._mvs_nhwc[0].data_ptr()
    assert torch.equal(_channels_last(feats[0]).permute(0, 3, 1, 2), feats[0])


@pytest.mark.gpu
def test_fpn_encoder_decoder_full_size_vs_torch_on_gpu():
    """Encoder + decoder chained at BASELINE configs[1] size (5 views, 1152x1536) against the oracle's torch ops on the GPU."""
    from oracle import ref_fpn
    dev = torch.device("cuda:0")
    enc, dec = build_encoder(seed=31).to(dev), build_decoder(seed=32).to(dev)
    x = torch.randn(5, 3, 1152, 1536, generator=torch.Generator().manual_seed(33)).to(dev)
    feats = enc(x)
    outs = dec(*feats)
    with torch.no_grad():
        wf = ref_fpn.fpn_encoder_forward({k: v.detach() for k, v in enc.state_dict().items()}, x)
        wo = ref_fpn.fpn_decoder_forward({k: v.detach() for k, v in dec.state_dict().items()}, *wf)
    for i, (o, ww) in enumerate(zip(list(feats) + list(outs), list(wf) + list(wo))):
        assert o.shape == ww.shape
        err = float((o - ww).abs().max() / ww.abs().max())
        assert err < 1e-4, (i, err)


@pytest.mark.gpu
def test_fpn_training_mode_vs_reference_gradients():
    """FPNEncoder + FPNDecoder in TRAINING mode on the HIP path (batch-statistics BatchNorm; convolutions, their data and weight gradients
    as split-form GEMMs; bilinear upsampling and its adjoint) against outputs, EVERY parameter gradient, the input gradient and the updated
    running statistics of the reference's own modules (tests/golden/fpn_train.npz, oracle/gen_golden.py::gen_fpn_train)."""
    import json
    import os
    import mvsformer_amd as m
    from oracle.weights import make_state_dict
    dev = torch.device("cuda:0")
    g = load_golden("fpn_train.npz")
    shapes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fpn_shapes.json")))
    enc, dec = m.FPNEncoder([8, 16, 32, 64]), m.FPNDecoder([8, 16, 32, 64])
    enc.load_state_dict(make_state_dict(shapes["encoder"], int(g["seeds"][0])), strict=True)
    dec.load_state_dict(make_state_dict(shapes["decoder"], int(g["seeds"][1])), strict=True)
    enc, dec = enc.to(dev).train(), dec.to(dev).train()
    x = torch.from_numpy(g["x"].astype(np.float32)).to(dev).requires_grad_(True)
    feats = enc(x)
    outs = dec(*feats)
    gen = torch.Generator().manual_seed(int(g["seeds"][2]))
    torch.randn(2, 3, 32, 40, generator=gen)               # the generator state after the image, as in gen_fpn_train
    R = [torch.randn(o.shape, generator=gen).to(dev) for o in outs]
    loss = sum((o * r).sum() for o, r in zip(outs, R))
    loss.backward()
    torch.cuda.synchronize()

    def rel(a, b):
        b = torch.from_numpy(np.asarray(b)).to(torch.float64)
        return (a.detach().double().cpu() - b).abs().max().item() / max(1e-9, b.abs().max().item())
    for i, (f, o) in enumerate(zip(feats, outs)):
        assert rel(f, g["feat%d" % i]) < 2e-5, ("feat", i, rel(f, g["feat%d" % i]))
        assert rel(o, g["out%d" % i]) < 2e-5, ("out", i, rel(o, g["out%d" % i]))
    assert rel(x.grad, g["dx"]) < 2e-4, rel(x.grad, g["dx"])
    worst = ("", 0.0)
    for tag, mod in (("enc", enc), ("dec", dec)):
        for k, p in mod.named_parameters():
            want = g["%s.grad.%s" % (tag, k)]
            e = rel(p.grad, want)
            # a conv bias in front of a batch-statistics BatchNorm has an exactly-zero gradient in exact arithmetic: pure rounding noise
            tol = 2e-4 if np.abs(want).max() > 1e-3 else None
            if tol is None:
                assert p.grad.abs().max().item() < 1e-3, (tag, k)
                continue
            worst = max(worst, ("%s.%s" % (tag, k), e), key=lambda t: t[1])
            assert e < tol, (tag, k, e)
        for k, b in mod.named_buffers():
            if b.dtype.is_floating_point:
                assert rel(b, g["%s.buf.%s" % (tag, k)]) < 2e-5, (tag, k)
    print("worst parameter gradient", worst)


# ------------------------------------------------------------------------------------------ FPNDecoderV2 (models/module.py:273-302)
V2_INPUTS = ("conv01", "conv11", "conv21", "conv31", "vit1", "vit2", "vit3")


def build_decoder_v2(sd=None, seed=0):
    from mvsformer_amd import FPNDecoderV2
    from oracle import ref_fpn
    torch.manual_seed(seed)
    dec = FPNDecoderV2([8, 16, 32, 64])
    if sd is None:
        ref_fpn.randomize_bn(dec, seed + 1)
    else:
        missing, unexpected = dec.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return dec.eval()


def v2_case(seed, N, h, w):
    from oracle import ref_fpn
    g = torch.Generator().manual_seed(seed + 100)
    return ref_fpn.make_case(seed, N, h, w) + tuple(torch.randn(N, c, h * s, w * s, generator=g) for c, s in ((64, 1), (32, 2), (16, 4)))


def test_fpn_decoder_v2_is_checkpoint_compatible_and_has_no_cpu_path():
    """CPU: the reference's parameter names (the golden state_dict loads strictly up to num_batches_tracked); CPU tensors raise in both modes."""
    g = load_golden("fpn_decoder_v2.npz")
    dec = build_decoder_v2({k[3:]: t(v.astype(np.float32)) for k, v in g.items() if k.startswith("sd.")})
    assert sorted(k for k in dec.state_dict() if not k.endswith("num_batches_tracked")) == sorted(k[3:] for k in g if k.startswith("sd."))
    from mvsformer_amd._lib import MvsHipError
    with pytest.raises(MvsHipError):
        dec.train()(*[t(g[k]) for k in V2_INPUTS])
    with pytest.raises(MvsHipError):
        dec.eval()(*[t(g[k]) for k in V2_INPUTS])


@pytest.mark.gpu
def test_fpn_decoder_v2_training_mode_vs_reference_gradients():
    """``FPNDecoderV2`` in TRAINING mode (batch-statistics BatchNorm + Swish, ConvTranspose2d + BatchNorm + ReLU + the encoder map) against
    the real module in train() (tests/golden/fpn_decoder_v2_train.npz): outputs, loss, the gradient of all seven inputs, every parameter
    gradient (sampled + norm) and the updated running statistics."""
    import mvsformer_amd as m
    from oracle.weights import make_state_dict
    import json, os
    g = load_golden("fpn_decoder_v2_train.npz")
    shapes = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fpn_v2_shapes.json")))
    dev = torch.device("cuda:0")
    dec = m.FPNDecoderV2([8, 16, 32, 64])
    dec.load_state_dict(make_state_dict(shapes, int(g["seeds"][0])), strict=True)
    dec = dec.to(dev).train()
    names = ("conv01", "conv11", "conv21", "conv31", "vit1", "vit2", "vit3")
    ins = {k: t(g["in." + k].astype(np.float32), dev).requires_grad_(True) for k in names}
    outs = dec(*[ins[k] for k in names])
    gen = torch.Generator().manual_seed(int(g["seeds"][1]))
    for k in names:
        torch.randn(ins[k].shape, generator=gen)
    loss = sum((o * torch.randn(o.shape, generator=gen).to(dev)).sum() for o in outs)
    loss.backward()
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert scale_err(o.detach().cpu(), g["out%d" % i]) < 2e-5, i
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    for k in names:
        assert scale_err(ins[k].grad.cpu(), g["din." + k]) < 2e-4, k
    for k, p in dec.named_parameters():
        want, idx = t(g["grad." + k]), torch.from_numpy(g["idx." + k].astype(np.int64))
        got = p.grad.detach().cpu().reshape(-1)[idx]
        if want.abs().max() < 1e-3:                         # conv bias in front of a batch-statistics BatchNorm: zero up to rounding noise
            assert got.abs().max() < 1e-3, k
            continue
        assert (got - want).abs().max() < 2e-4 * want.abs().max(), k
        assert abs(float(p.grad.double().norm()) - float(g["norm." + k])) < 2e-4 * float(g["norm." + k]), k
    for k, b in dec.named_buffers():
        if b.dtype.is_floating_point:
            assert (b.cpu() - t(g["buf." + k])).abs().max() < 1e-5, k


@pytest.mark.gpu
def test_fpn_decoder_v2_vs_golden():
    """Against the outputs of the reference's own FPNDecoderV2 (oracle/gen_golden.py::gen_fpn_decoder_v2): split-form GEMMs are
    fp32-equivalent, four layers deep -> 2e-5 of each map's scale."""
    g = load_golden("fpn_decoder_v2.npz")
    dev = torch.device("cuda:0")
    dec = build_decoder_v2({k[3:]: t(v.astype(np.float32)) for k, v in g.items() if k.startswith("sd.")}).to(dev)
    outs = dec(*[t(g[k], dev) for k in V2_INPUTS])
    assert len(outs) == 4
    for i, o in enumerate(outs, start=1):
        assert tuple(o.shape) == g["out%d" % i].shape
        assert o.permute(0, 2, 3, 1).is_contiguous()            # channel-last memory, the reference's logical shape
        assert scale_err(o.cpu(), g["out%d" % i]) < TOL, (i, scale_err(o.cpu(), g["out%d" % i]))


@pytest.mark.gpu
@pytest.mark.parametrize("N,h,w", [(2, 7, 9), (1, 1, 1), (3, 2, 17)])
def test_fpn_decoder_v2_vs_oracle(N, h, w):
    """Other shapes (ragged tiles, several images, the 1x1 coarsest level) against the CPU oracle (oracle/ref_fpn.py)."""
    from oracle import ref_fpn
    dec = build_decoder_v2(seed=3)
    ins = v2_case(4, N, h, w)
    want = ref_fpn.fpn_decoder_v2_forward({k: v.detach() for k, v in dec.state_dict().items()}, *ins)
    dec = dec.to("cuda:0")
    outs = dec(*[f.to("cuda:0") for f in ins])
    for i, (o, ww) in enumerate(zip(outs, want)):
        assert o.shape == ww.shape
        assert scale_err(o.cpu(), ww) < TOL, (i, scale_err(o.cpu(), ww))
