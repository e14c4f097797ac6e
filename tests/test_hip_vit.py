"""The DINO ViT-small branch (csrc/vit.hip, mvsformer_amd/vit.py) on the GPU against the reference's own outputs
(tests/golden/vit_small.npz, made by oracle/gen_golden.py::gen_vit from the real ``vits.vit_small`` + ``VITDecoderStage4Single``) and the
primitive kernels against plain torch (CPU, fp64 where it matters)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(got, want):
    return (got.double().cpu() - want.double()).abs().max().item() / max(1e-12, want.double().abs().max().item())


def test_vit_branch_vs_reference_golden(dev):
    import mvsformer_amd as m
    from mvsformer_amd import vit as V
    from oracle.weights import load_vit_shapes, make_vit_state_dict
    g = load_golden("vit_small.npz")
    net = m.vit_small(patch_size=16, qk_scale="default")
    dec = m.VITDecoderStage4Single(dict(out_ch=64, vit_ch=384, att_fusion=True, nhead=6))
    net.load_state_dict(make_vit_state_dict(load_vit_shapes("vit_small"), int(g["seeds"][0])), strict=True)
    dec.load_state_dict(make_vit_state_dict(load_vit_shapes("vit_decoder"), int(g["seeds"][1])), strict=True)
    net, dec = net.to(dev).eval(), dec.to(dev).eval()
    img = torch.from_numpy(g["img"].astype(np.float32)).to(dev)
    out = V.vit_branch(net, dec, img)
    torch.cuda.synchronize()
    # fp32-equivalent arithmetic through 12 blocks: 1e-4 of the output scale, the resize to a few ulps
    for k, tol in (("vit_imgs", 2e-6), ("vit_feat", 1e-4), ("att_cls", 1e-4), ("vit_out", 1e-4)):
        assert _rel(out[k], torch.from_numpy(g[k])) < tol, (k, _rel(out[k], torch.from_numpy(g[k])))
    assert out["vit_out"].shape == (1, 64, 32, 40)
    with pytest.raises(Exception):
        net.train()(img)                                   # eval-only on the HIP path: fails loudly


@pytest.mark.parametrize("shape", [(1, 1, 70, 130, 100, False), (2, 3, 33, 64, 17, True), (1, 2, 129, 65, 384, False)])
def test_gemm_x3_vs_fp64(dev, shape):
    from mvsformer_amd import ops
    nb1, nb2, M, N, K, kn = shape
    gen = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(nb1, nb2, M, K, generator=gen)
    B = torch.randn(nb1, nb2, K, N, generator=gen) if kn else torch.randn(nb1, nb2, N, K, generator=gen)
    scale, shift = torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen)
    res = torch.randn(nb1, nb2, M, N, generator=gen)
    C = torch.empty(nb1, nb2, M, N, device=dev)
    ops.gemm_x3(A.to(dev), B.to(dev), C, M, N, K, K, N if kn else K, N, nb1=nb1, nb2=nb2, sA=(nb2 * M * K, M * K), sB=(nb2 * N * K, N * K),
                sC=(nb2 * M * N, M * N), b_kn=kn, alpha=0.5, scale=scale.to(dev), shift=shift.to(dev), act=1, res=res.to(dev))
    want = 0.5 * (A.double() @ (B.double() if kn else B.double().transpose(-1, -2))) * scale.double() + shift.double()
    want = F.gelu(want) + res.double()
    fp32 = F.gelu((0.5 * (A @ (B if kn else B.transpose(-1, -2)))) * scale + shift) + res
    err, err32 = _rel(C, want), _rel(fp32, want)
    assert err < 3 * err32 + 2e-7, (err, err32)            # the split form's contract: as close to fp64 as an fp32 product chain


def test_gemm_x3_implicit_convs(dev):
    from mvsformer_amd import ops
    from mvsformer_amd.vit import _conv3_matrix, _convT_matrices
    gen = torch.Generator().manual_seed(5)
    B, h, w, cin, cout = 2, 7, 9, 14, 24
    x = torch.randn(B, cin, h, w, generator=gen)
    w3 = torch.randn(cout, cin, 3, 3, generator=gen) * 0.2
    cp = 16
    xcl = torch.zeros(B, h, w, cp)
    xcl[..., :cin] = x.permute(0, 2, 3, 1)
    out = torch.empty(B, h * w, cout, device=dev)
    ops.gemm_x3(xcl.to(dev), _conv3_matrix(w3.to(dev), cp), out, h * w, cout, 9 * cp, 0, 9 * cp, cout, nb1=B, sA=(h * w * cp, 0), sC=(h * w * cout, 0),
                a_mode=1, H=h, W=w, Cp=cp)
    want = F.conv2d(x.double(), w3.double(), padding=1).permute(0, 2, 3, 1).reshape(B, h * w, cout)
    assert _rel(out, want) < 1e-6
    wt = torch.randn(cp, cout, 4, 4, generator=gen) * 0.2
    tmp = torch.empty(B, 4, h * w, cout, device=dev)
    ops.gemm_x3(xcl.to(dev), _convT_matrices(wt.to(dev)), tmp, h * w, cout, 4 * cp, 0, 4 * cp, cout, nb1=B, nb2=4, sA=(h * w * cp, 0), sB=(0, cout * 4 * cp),
                sC=(4 * h * w * cout, h * w * cout), a_mode=2, H=h, W=w, Cp=cp)
    got = tmp.view(B, 2, 2, h, w, cout).permute(0, 3, 1, 4, 2, 5).reshape(B, 2 * h, 2 * w, cout)
    want = F.conv_transpose2d(xcl.permute(0, 3, 1, 2).double(), wt.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert _rel(got, want) < 1e-6


def test_layernorm_softmax_bicubic(dev):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(37, 384, generator=gen) * 3 + 1
    g, b = torch.rand(384, generator=gen) + 0.5, torch.randn(384, generator=gen)
    assert _rel(ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-6), F.layer_norm(x.double(), (384,), g.double(), b.double(), 1e-6)) < 2e-6
    s = torch.randn(11, 1729, generator=gen) * 4
    assert _rel(ops.softmax_rows_(s.clone().to(dev), 0.125), torch.softmax(s.double() * 0.125, -1)) < 2e-6
    img = torch.randn(2, 3, 50, 62, generator=gen)
    want = F.interpolate(img, (25, 31), mode="bicubic", align_corners=False)
    assert _rel(ops.bicubic_resize(img.to(dev), 25, 31, 50 / 25, 62 / 31), want) < 2e-6
    tab = torch.randn(8, 14, 14, generator=gen)
    sh, sw = (8 + 0.1) / 14, (10 + 0.1) / 14
    want = F.interpolate(tab[None], scale_factor=(sh, sw), mode="bicubic", align_corners=False)[0]
    assert want.shape[-2:] == (8, 10)
    assert _rel(ops.bicubic_resize(tab.to(dev), 8, 10, 1 / sh, 1 / sw), want) < 2e-6


@pytest.mark.parametrize("shape", [(1, 1, 50), (2, 6, 197), (1, 3, 1729), (3, 2, 64)])
def test_attention_x3_flash_vs_fp64(dev, shape):
    """Flash-form attention (online softmax, no N x N matrix) against softmax(Q K^T / 8) V in fp64, and against the materialized
    split-form path it replaces in blocks 0..10 (vision_transformer.py:60-76 is the reference's Attention.forward)."""
    from mvsformer_amd import ops
    B, NH, N = shape
    hd, C = 64, NH * 64
    gen = torch.Generator().manual_seed(N + 3 * B)
    qkv = torch.randn(B, N, 3 * C, generator=gen) * 1.7
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].reshape(B, N, NH, hd).permute(0, 2, 1, 3).double() for i in range(3))
    want = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    fp32 = (torch.softmax(q.float() @ k.float().transpose(-1, -2) * hd ** -0.5, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, C)
    d = qkv.to(dev)
    vt = ops.attention_vt(d, NH)
    assert vt.shape == (B, NH, hd, (N + 3) // 4 * 4)
    got = ops.attention_x3(d, vt, NH, hd ** -0.5)
    torch.cuda.synchronize()
    err, err32 = _rel(got, want), _rel(fp32, want)
    assert err < 3 * err32 + 5e-7, (err, err32)
    with pytest.raises(Exception):
        ops.attention_x3(d[:, :, :96].contiguous(), vt[:, :1, :32].contiguous(), 1, 0.1)      # head dimension 32: refused loudly, no silent fallback


# ------------------------------------------------------------------------------------------------------------ pre-split ("packed") operands
def test_x3p_pack_roundtrip_and_layout(dev):
    """pack -> unpack is EXACT (x = h + m + l), rows beyond R are zeros, and the layout is the documented one (include/mvs_hip.h):
    piece (rt, ks, term) at ((rt*K/32 + ks)*3 + term) KiB, lane = ((k%32)/8)*16 + r%16, element k%8."""
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(37, 96, generator=gen) * torch.tensor(10.0) ** torch.randint(-20, 20, (37, 1), generator=gen).float()
    p = ops.x3p_pack(x.to(dev), rows_alloc=48)
    assert torch.equal(ops.x3p_unpack(p).cpu(), x)
    raw = p.buf.cpu().view(torch.bfloat16).float().reshape(3, 3, 3, 64, 8)          # [rt][ks][term][lane][8]
    r, k = 21, 77
    lane = ((k % 32) // 8) * 16 + r % 16
    terms = raw[r // 16, k // 32, :, lane, k % 8]
    h = x[r, k].bfloat16().float()
    m = (x[r, k] - h).bfloat16().float()
    assert terms[0] == h and terms[1] == m and terms[2] == (x[r, k] - h - m).bfloat16().float()
    assert raw[2, :, :, 5:16].abs().max() == 0 and raw[2, :, :, 21:32].abs().max() == 0   # rows 37..47: zeros


@pytest.mark.parametrize("shape", [(200, 132, 96, 1), (128, 128, 32, 0), (391, 384, 384, 1), (8645, 64, 160, 0)])
def test_gemm_x3p_vs_fp64(dev, shape):
    """The pre-split GEMM (LDS-DMA fed, no split in the main loop) against fp64 with an fp32 product chain's own error as the yardstick -
    the same contract as test_gemm_x3_vs_fp64 - and bit-equal to the split-on-the-fly kernel's arithmetic is NOT required (the MFMA
    order within a K step is the same, the tile shapes differ only in who computes what)."""
    from mvsformer_amd import ops
    M, N, K, act = shape
    gen = torch.Generator().manual_seed(M + N)
    A, B = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen)
    scale, shift, res = torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
    Ap, Bp = ops.x3p_pack(A.to(dev)), ops.x3p_pack(B.to(dev))
    C = torch.full((M, N), float("nan"), device=dev)
    outp = ops.Packed(M, N, dev, zero=True) if N % 32 == 0 else None
    ops.gemm_x3p(Ap, Bp, N, C=C, scale=scale.to(dev), shift=shift.to(dev), act=act, res=res.to(dev), out=outp)
    want = (A.double() @ B.double().t()) * scale.double() + shift.double()
    want = (F.gelu(want) if act else want) + res.double()
    v32 = (A @ B.t()) * scale + shift
    fp32 = (F.gelu(v32) if act else v32) + res
    err, err32 = _rel(C, want), _rel(fp32, want)
    assert err < 3 * err32 + 2e-7, (err, err32)
    if outp is not None:
        assert torch.equal(ops.x3p_unpack(outp), C)         # the packed copy of the output is the same numbers, exactly
        only = ops.Packed(M, N, dev, zero=True)
        ops.gemm_x3p(Ap, Bp, N, scale=scale.to(dev), shift=shift.to(dev), act=act, out=only)     # packed output alone, no residual
        ops.gemm_x3p(Ap, Bp, N, C=C, scale=scale.to(dev), shift=shift.to(dev), act=act)
        assert torch.equal(ops.x3p_unpack(only), C)


def test_layernorm_x3p(dev):
    from mvsformer_amd import ops
    gen = torch.Generator().manual_seed(4)
    Np, N, C = 64, 37, 384
    x = torch.randn(2 * Np, C, generator=gen) * 3 + 1
    g, b = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    got = ops.x3p_unpack(ops.layernorm_x3p(x.to(dev), g.to(dev), b.to(dev), 1e-6, Np, N)).cpu().reshape(2, Np, C)
    want = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-6).reshape(2, Np, C)
    assert _rel(got[:, :N], want[:, :N]) < 2e-6
    assert got[:, N:].abs().max() == 0                     # padding rows: zeros


@pytest.mark.parametrize("shape", [(1, 2, 50), (2, 6, 197), (1, 6, 1729), (3, 2, 64), (2, 2, 5), (1, 4, 33)])
def test_attention_x3p_vs_fp64(dev, shape):
    """qkv GEMM -> packed Q / K / V^T -> flash attention -> packed output, and the CLS row, against fp64 attention on the fp32 qkv values
    the GEMM produced (the split-on-the-fly GEMM gives them as fp32)."""
    from mvsformer_amd import ops
    B, NH, N = shape
    hd, C = 64, NH * 64
    if C % 128:
        pytest.skip("C %% 128")
    Np = (N + 31) // 32 * 32
    gen = torch.Generator().manual_seed(N + 5 * B)
    x = torch.zeros(B, Np, C)
    x[:, :N] = torch.randn(B, N, C, generator=gen)
    W, bias = torch.randn(3 * C, C, generator=gen) * (1.7 / C ** 0.5), torch.randn(3 * C, generator=gen) * 0.3
    xp, wp = ops.x3p_pack(x.reshape(B * Np, C).to(dev)), ops.x3p_pack(W.to(dev))
    qkv_p = ops.gemm_x3p_qkv(xp, wp, bias.to(dev), B, Np, NH, hd ** -0.5)
    qkv = (x.double() @ W.double().t() + bias.double())[:, :N]
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].reshape(B, N, NH, hd).permute(0, 2, 1, 3) for i in range(3))
    # packed Q (scaled) and K against the fp64 products
    RT = Np // 16
    qraw = ops.Packed(B * NH * Np, 64, dev, rows_alloc=B * NH * Np)
    qraw.buf = qkv_p.q
    kraw = ops.Packed(B * NH * Np, 64, dev, rows_alloc=B * NH * Np)
    kraw.buf = qkv_p.k
    qg = ops.x3p_unpack(qraw).cpu().reshape(B, NH, Np, 64)[:, :, :N]
    kg = ops.x3p_unpack(kraw).cpu().reshape(B, NH, Np, 64)[:, :, :N]
    assert _rel(qg, q * (hd ** -0.5 * 1.4426950408889634)) < 2e-6 and _rel(kg, k) < 2e-6      # Q carries scale * log2(e): base-2 softmax
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1)
    want = (att @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    fp32 = (torch.softmax(q.float() @ k.float().transpose(-1, -2) * hd ** -0.5, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B, N, C)
    got = ops.x3p_unpack(ops.attention_x3p(qkv_p, N)).cpu().reshape(B, Np, C)[:, :N]
    err, err32 = _rel(got, want), _rel(fp32, want)
    assert err < 3 * err32 + 5e-7, (err, err32)
    cls = ops.cls_attention_x3p(qkv_p, N)
    assert _rel(cls, att[:, :, 0]) < 5e-6


def test_vit_packed_path_matches_split_on_the_fly_path(dev, monkeypatch):
    """The pre-split blocks (default) against the round-5 path (MVS_VIT_PACKED=0: operands split inside every GEMM block, fp32 activations
    between the layers): same arithmetic family, different tiling -> agreement far inside the 1e-4 the golden test allows."""
    import mvsformer_amd as m
    torch.manual_seed(3)
    net = m.vit_small(patch_size=16, qk_scale="default").to(dev).eval()
    img = torch.randn(2, 3, 80, 112, device=dev)
    tok, att = net.forward_with_cls_att(img)
    full_tok, full_att = net.forward_with_last_att(img)      # the whole attention matrix: the materialized path
    monkeypatch.setenv("MVS_VIT_PACKED", "0")
    net._cache = None
    tok0, att0 = net.forward_with_cls_att(img)
    assert tok.shape == tok0.shape == (2, 36, 384) and att.shape == att0.shape == (2, 6, 36)
    assert _rel(tok, tok0.cpu()) < 2e-5 and _rel(att, att0.cpu()) < 2e-5
    assert _rel(full_tok, tok0.cpu()) < 2e-5 and _rel(full_att[:, :, 0], att0.cpu()) < 2e-5


@pytest.mark.parametrize("cfg", [None, "8,128,4,2", "7,64,4,1", "7,128,8,1"])
def test_conv_x3p_implicit_convs(dev, cfg):
    """The implicit 3x3 / transposed 4x4 convolutions on packed channel-last maps (taps gathered per lane by LDS-DMA, out-of-image taps from the
    map's zero row) against torch's convolutions in fp64; ragged: 2 images of 7 x 9 pixels (row tiles straddle the images)."""
    import subprocess, sys, os
    if cfg is not None:                                    # the tile configuration is read once per process: check it in a child
        env = dict(os.environ, MVS_X3P_CFG=cfg)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "test_conv_x3p_implicit_convs and None"], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        return
    from mvsformer_amd import ops
    from mvsformer_amd.vit import _conv3_matrix, _convT_matrices
    gen = torch.Generator().manual_seed(5)
    B, h, w, cin, cout = 2, 7, 9, 44, 72
    cp = 64
    M = B * h * w
    x = torch.randn(B, cin, h, w, generator=gen)
    w3 = torch.randn(cout, cin, 3, 3, generator=gen) * 0.2
    xcl = torch.zeros(M, cp)
    xcl[:, :cin] = x.permute(0, 2, 3, 1).reshape(M, cin)
    ra = (M + 128) // 128 * 128
    xp = ops.x3p_pack(xcl.to(dev), ra)
    scale, shift = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
    mul = torch.randn(M, cout, generator=gen)
    out = torch.empty(M, cout, device=dev)
    ops.conv_x3p(xp, ops.x3p_pack(_conv3_matrix(w3.to(dev), cp)), 1, B, h, w, cout, C=out, scale=scale.to(dev), shift=shift.to(dev), act=2, mul=mul.to(dev))
    want = F.conv2d(x.double(), w3.double(), padding=1).permute(0, 2, 3, 1).reshape(M, cout) * scale.double() + shift.double()
    want = want * torch.sigmoid(want) * mul.double()
    assert _rel(out, want) < 2e-6
    # transposed convolution: the output rows interleave the four parity classes; fp32 and packed outputs
    cout2 = 64
    wt = torch.randn(cp, cout2, 4, 4, generator=gen) * 0.2
    wt[cin:] = 0
    wtp = ops.x3p_pack_classes(_convT_matrices(wt.to(dev)), 128)
    o2 = torch.empty(B, 2 * h, 2 * w, cout2, device=dev)
    o2p = ops.Packed(4 * M, cout2, dev, zero=True)
    ops.conv_x3p(xp, wtp, 2, B, h, w, cout2, C=o2.view(4 * M, cout2), act=3, out=o2p)
    want = torch.relu(F.conv_transpose2d(x.double(), wt[:cin].double(), stride=2, padding=1).permute(0, 2, 3, 1))
    assert _rel(o2, want) < 2e-6
    assert torch.equal(ops.x3p_unpack(o2p).reshape(B, 2 * h, 2 * w, cout2), o2)


def _mvsformer_p_args():
    return dict(fix=True, depth_type="ce", fusion_type="cnn", inverse_depth=True, attn_temp=2.0, base_ch=8, ndepths=[32, 16, 8, 4], feat_chs=[8, 16, 32, 64],
                depth_interals_ratio=[4.0, 2.67, 1.5, 1.0], multi_scale=False,
                vit_args=dict(twin=False, rescale=0.5, do_vit=True, patch_size=16, qk_scale="default", vit_arch="vit_small", vit_ch=384, out_ch=64,
                              att_fusion=True, nhead=6))


def test_dinomvsnet_images_to_depth_vs_reference_golden(dev):
    """The composed model, images -> depth map: ``mvsformer_amd.DINOMVSNet`` (FPN encoder -> DINO ViT branch -> FPN decoder -> 4-stage cascade,
    every piece a HIP kernel) against the REAL reference ``DINOMVSNet`` of configs/config_mvsformer-p.json run on the same images, cameras and
    seeded weights (tests/golden/dinomvsnet_e2e.npz, oracle/gen_golden.py::gen_end_to_end).  Depth within the north star's 1e-3 relative."""
    import mvsformer_amd as m
    from oracle.weights import load_model_shapes, make_model_state_dict
    g = load_golden("dinomvsnet_e2e.npz")
    net = m.DINOMVSNet(_mvsformer_p_args())
    sd = make_model_state_dict(load_model_shapes(), int(g["seed"]))
    net.load_state_dict(sd, strict=True)
    assert list(net.state_dict().keys()) == list(sd.keys())          # the reference's keys in the reference's order
    net = net.to(dev).eval()
    imgs = torch.from_numpy(g["imgs"].astype(np.float32)).to(dev)
    proj = {"stage%d" % i: torch.from_numpy(g["proj_stage%d" % i]).to(dev) for i in range(1, 5)}
    dv = torch.from_numpy(g["depth_range"]).to(dev)
    feats = net.extract_features(imgs)
    f1 = torch.from_numpy(g["features_stage1"])
    assert _rel(feats["stage1"], f1) < 1e-4
    out = net(imgs, proj, dv, tmp=[float(t) for t in g["tmps"]])
    torch.cuda.synchronize()
    for i in range(1, 5):
        want = torch.from_numpy(g["s%d_depth" % i])
        rel = ((out["stage%d" % i]["depth"].cpu() - want).abs() / want.abs()).max().item()
        assert rel < 1e-3, (i, rel)
    want = torch.from_numpy(g["refined_depth"])
    rel = ((out["refined_depth"].cpu() - want).abs() / want.abs()).max().item()
    assert rel < 1e-3, rel
    assert (out["photometric_confidence"].cpu() - torch.from_numpy(g["photometric_confidence"])).abs().max() < 2e-3


def test_vit_decoder_training_mode_vs_reference_gradients(dev):
    """``VITDecoderStage4Single`` in TRAINING mode on the HIP path (batch-statistics BatchNorm, Swish / GELU, the gated product, both transposed
    convolutions; forward and backward are HIP kernels) against the real reference module in train(): output, loss, input gradients, every
    parameter gradient (the big tensors by a fixed 4096-element sample + their norm) and the updated running statistics."""
    import mvsformer_amd as m
    from oracle.weights import load_vit_shapes, make_vit_state_dict
    g = load_golden("vit_decoder_train.npz")
    dec = m.VITDecoderStage4Single(dict(out_ch=64, vit_ch=384, att_fusion=True, nhead=6))
    dec.load_state_dict(make_vit_state_dict(load_vit_shapes("vit_decoder"), int(g["seeds"][0])), strict=True)
    dec = dec.to(dev).train()
    feat = torch.from_numpy(g["feat"].astype(np.float32)).to(dev).requires_grad_(True)
    att = torch.from_numpy(g["att"].astype(np.float32)).to(dev).requires_grad_(True)
    out = dec(feat, att)
    gen = torch.Generator().manual_seed(int(g["seeds"][1]))
    torch.randn(2, 384, 8, 10, generator=gen), torch.rand(2, 6, 8, 10, generator=gen)
    R = torch.randn(out.shape, generator=gen).to(dev)
    loss = (out * R).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert _rel(out, torch.from_numpy(g["out"])) < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert _rel(feat.grad, torch.from_numpy(g["dfeat"])) < 2e-4 and _rel(att.grad, torch.from_numpy(g["datt"])) < 2e-4
    worst = 0.0
    for k, p in dec.named_parameters():
        want, idx = torch.from_numpy(g["grad." + k]), torch.from_numpy(g["idx." + k])
        got = p.grad.detach().cpu().reshape(-1)[idx]
        if want.abs().max() < 1e-2:                        # a conv bias in front of a batch-statistics BatchNorm: exactly zero in exact arithmetic,
            assert got.abs().max().item() < 1e-2, k        # rounding noise on both sides (the neighbouring gradients are ~1e2)
            continue
        e = (got - want).abs().max().item() / max(1e-12, want.abs().max().item())
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
        assert abs(float(p.grad.double().norm()) - float(g["norm." + k])) < 2e-4 * float(g["norm." + k]) + 1e-7, k
    for k, b in dec.named_buffers():
        if b.dtype.is_floating_point:
            assert (b.cpu() - torch.from_numpy(g["buf." + k])).abs().max() < 1e-5, k
    print("worst parameter-gradient error %.2e" % worst)


def test_mvsformer_p_training_step_vs_reference(dev):
    """One training step of the whole MVSFormer-P model on the HIP path - frozen DINO ViT -> VITDecoderStage4Single (train) -> FPN encoder /
    decoder (train) -> four StageNets (train) -> ce_loss_stage4 -> backward -> FusedAdamW - against the REAL reference ``DINOMVSNet`` +
    ``ce_loss_stage4`` run on the same images, cameras, ground truth and seeded weights (tests/golden/train_step_mvsformer_p.npz, fp32, 3 views of
    256 x 320): the four stage losses and a 512-element sample + the norm of every one of the 231 gradients.

    A training-mode stage picks the arg-max hypothesis (mvsformer_model.py:118-121) and normalizes with BATCH statistics, so the free-running
    cascade is chaotic in the last ulp: a pixel whose two best logits tie (0.08 % at stage 1) moves a whole bin, its 4 x 4 children get other
    hypotheses, the next stage's batch statistics move, ... (mismatching pixels 0.08 % -> 3 % -> 22 % -> 43 % over the stages, while each stage
    fed the SAME inputs agrees with the oracle to 6e-5 in the logits and in every arg-max).  So: (1) the free-running ``forward`` is checked
    at stage 1 (depth as a mismatch fraction, loss) and for finite losses; (2) the step itself is compared TEACHER-FORCED - every stage gets the
    hypotheses the reference's own previous stage produced (``schedule_inverse_range`` of the stored reference depths; they are detached in the
    model anyway), which is the reference's computation graph exactly - losses to 1e-5.  The GRADIENTS of this model are themselves
    ill-conditioned at a random-weight operating point: the REAL reference StageNet's own gradients move by 0.7 % (median over its tensors,
    relative L2) and up to 15 % (the visibility CNN's last bias) when its stage-1 features are perturbed by 1e-5 relative (measured with the
    reference classes on the CPU; the oracle restatement equals them bit for bit on identical inputs) - and our FPN + ViT features differ from
    the reference's by about that much.  Hence (2) holds the sampled gradients to 3e-2 (median per sub-module) / 0.25 (any tensor), and (3)
    pins the backward where it CAN be pinned: stages 1 and 2 against the oracle under torch autograd on IDENTICAL inputs, every parameter
    gradient and d loss / d features to 2e-4 / 5e-3 relative L2."""
    import mvsformer_amd as m
    from mvsformer_amd import losses, synth
    from mvsformer_amd.optim import FusedAdamW
    from oracle.weights import load_model_shapes, make_model_state_dict
    g = load_golden("train_step_mvsformer_p.npz")
    net = m.DINOMVSNet(_mvsformer_p_args())
    net.load_state_dict(make_model_state_dict(load_model_shapes(), int(g["seed"])), strict=True)
    net = net.to(dev).train()
    imgs = torch.from_numpy(g["imgs"].astype(np.float32)).to(dev)
    V, H, W = imgs.shape[1], imgs.shape[3], imgs.shape[4]
    proj = {"stage%d" % i: torch.from_numpy(g["proj_stage%d" % i]).to(dev) for i in range(1, 5)}
    dv = torch.from_numpy(g["depth_range"]).to(dev)
    tmps = [float(t) for t in g["tmps"]]
    scene = synth.make_scene(V, H, W, int(g["scene_seed"]))
    gts = {"stage%d" % (i + 1): synth.plane_depth(scene, s).to(torch.float32).unsqueeze(0).to(dev) for i, s in enumerate((8, 4, 2, 1))}
    masks = {k: torch.ones_like(v) for k, v in gts.items()}
    dlossw = [float(w) for w in g["dlossw"]]
    state0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # (1) free-running forward
    out = net(imgs, proj, dv, tmp=tmps)
    ls = losses.ce_loss_stage4(out, gts, masks, dlossw, inverse_depth=True)
    want = torch.from_numpy(g["s1_depth"])
    assert ((out["stage1"]["depth"].detach().cpu() - want).abs() > 1e-3 * want.abs()).float().mean().item() < 3e-3
    assert abs(float(ls["stage1"].detach()) - float(g["losses"][0])) < 1e-3 * float(g["losses"][0])
    assert all(torch.isfinite(v.detach()).all() for v in ls.values())
    del out, ls
    net.load_state_dict(state0, strict=True)               # the running statistics took one update: back to the golden's starting point
    # (2) the teacher-forced step
    opt = FusedAdamW([p for p in net.parameters()], lr=1e-3)
    feats = net.extract_features(imgs)
    outs, hyp, prev = {}, None, None
    for i in range(4):
        f = feats["stage%d" % (i + 1)]
        if i == 0:
            hyp = m.init_inverse_range(dv, net.ndepths[0], dev, torch.float32, f.shape[-2], f.shape[-1])
        else:
            hyp = m.schedule_inverse_range(prev, hyp, net.ndepths[i], net.depth_interals_ratio[i], f.shape[-2], f.shape[-1])
        outs["stage%d" % (i + 1)] = net.fusions[i](f, proj["stage%d" % (i + 1)], hyp, tmp=tmps)
        prev = torch.from_numpy(g["s%d_depth" % (i + 1)]).to(dev)         # the REFERENCE's depth of this stage schedules the next one
        got = outs["stage%d" % (i + 1)]["depth"].detach().cpu()
        assert ((got - prev.cpu()).abs() > 1e-3 * prev.cpu().abs()).float().mean().item() < 3e-3, i
    ls = losses.ce_loss_stage4(outs, gts, masks, dlossw, inverse_depth=True)
    sum(ls.values()).backward()
    torch.cuda.synchronize()
    print("stage losses: " + ", ".join("%.6f (reference %.6f)" % (float(ls["stage%d" % i].detach()), float(g["losses"][i - 1])) for i in range(1, 5)))
    for i in range(1, 5):
        assert abs(float(ls["stage%d" % i].detach()) - float(g["losses"][i - 1])) < 1e-5 * float(g["losses"][i - 1]), (i, float(ls["stage%d" % i]), float(g["losses"][i - 1]))
    errs, n = [], 0
    for k, p in net.named_parameters():
        if "grad." + k not in g:
            assert p.grad is None and k.startswith("vit."), k             # the frozen ViT: no gradients on either side
            continue
        want, idx = torch.from_numpy(g["grad." + k]), torch.from_numpy(g["idx." + k].astype(np.int64))
        got = p.grad.detach().cpu().reshape(-1)[idx]
        n += 1
        if want.abs().max() < 1e-3 * max(1.0, float(g["norm." + k])) and float(g["norm." + k]) < 1e-2:
            assert got.abs().max().item() < 1e-2, k        # conv biases in front of batch-statistics BatchNorm: zero up to rounding noise
            continue
        e = float((got - want).double().norm()) / max(1e-12, float(want.double().norm()))
        errs.append((e, k))
    assert n == 231
    errs.sort(reverse=True)
    print("relative L2 gradient errors over the samples, ten worst: " + ", ".join("%s %.1e" % (k, e) for e, k in errs[:10]))
    print("by sub-module (median / max): " + "; ".join("%s %.1e / %.1e" % (pre, sorted(e for e, k in errs if k.startswith(pre))[len([1 for e, k in errs if k.startswith(pre)]) // 2],
                                                                             max(e for e, k in errs if k.startswith(pre))) for pre in ("encoder", "decoder.", "decoder_vit", "fusions.0", "fusions.1", "fusions.2", "fusions.3")))
    assert errs[0][0] < 0.25, errs[0]
    for pre in ("encoder", "decoder.", "decoder_vit", "fusions.0", "fusions.1", "fusions.2", "fusions.3"):
        sub = sorted(e for e, k in errs if k.startswith(pre))
        assert sub[len(sub) // 2] < 3e-2, (pre, sub[len(sub) // 2])
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    moved = sum(int(not torch.equal(p.detach(), before[k])) for k, p in net.named_parameters() if p.grad is not None)
    assert moved == 231                                    # every trained tensor took its AdamW update; the frozen ViT did not
    assert all(torch.equal(p.detach(), before[k]) for k, p in net.named_parameters() if k.startswith("vit."))
    # (3) the backward on IDENTICAL inputs: HIP StageNet + HIP loss against the oracle (torch autograd on the CPU) for stages 1 and 2
    from oracle import ref_losses, ref_torch
    sd = make_model_state_dict(load_model_shapes(), int(g["seed"]))
    net.load_state_dict(state0, strict=True)
    hyp1 = None
    for i, tol in ((1, 2e-4), (2, 5e-3)):
        k = "stage%d" % i
        f = feats[k].detach().contiguous().float().clone().requires_grad_(True)
        if i == 1:
            hyp = hyp1 = m.init_inverse_range(dv, net.ndepths[0], dev, torch.float32, f.shape[-2], f.shape[-1])
        else:
            hyp = m.schedule_inverse_range(torch.from_numpy(g["s1_depth"]).to(dev), hyp1, net.ndepths[1], net.depth_interals_ratio[1], f.shape[-2], f.shape[-1])
        st = net.fusions[i - 1]
        st.zero_grad()
        o = st(f, proj[k], hyp, tmp=5.0)
        four = ("stage1", "stage2", "stage3", "stage4")
        l_hip = losses.ce_loss_stage4({q: o for q in four}, {q: gts[k] for q in four}, {q: masks[k] for q in four}, [1.0, 0.0, 0.0, 0.0], inverse_depth=True)["stage1"]
        l_hip.backward()
        sub = {q[len("fusions.%d." % (i - 1)):]: v.clone() for q, v in sd.items() if q.startswith("fusions.%d." % (i - 1))}
        params = {q: v.requires_grad_(True) for q, v in sub.items() if v.dtype.is_floating_point and "running" not in q}
        fc = f.detach().cpu().clone().requires_grad_(True)
        ref = ref_torch.stage_forward(fc, proj[k].cpu(), hyp.cpu(), sub, G=8, ndepth=net.ndepths[i - 1], model_th=8, tmp=5.0, training=True)
        l_ref = ref_losses.ce_loss_stage(ref["prob_volume_pre"], hyp.cpu(), gts[k].cpu(), masks[k].cpu(), inverse_depth=True)
        l_ref.backward()
        assert abs(float(l_hip.detach()) - float(l_ref.detach())) < 1e-5 * float(l_ref.detach())
        assert float((f.grad.cpu() - fc.grad).double().norm() / fc.grad.double().norm()) < tol, k
        for name, p in st.named_parameters():
            a, b = p.grad.detach().cpu().double(), params[name].grad.double()
            if float(b.norm()) < 1e-6 * b.numel() ** 0.5:
                continue                                    # (conv biases in front of a batch-statistics BatchNorm)
            assert float((a - b).norm() / b.norm()) < tol, (k, name, float((a - b).norm() / b.norm()))


def test_dinomvsnet_batch_of_two_equals_two_single_samples(dev):
    """``DINOMVSNet`` in eval batches the B x V views through the 2-D networks: a batch of two samples (different images, same cameras) must give
    each sample the depth map it gets alone (eval BatchNorm: no coupling between images), bit for bit."""
    import mvsformer_amd as m
    from mvsformer_amd import synth
    torch.manual_seed(1)
    net = m.DINOMVSNet(_mvsformer_p_args()).eval()
    m.cascade.randomize_bn_(net, seed=2)
    net = net.to(dev)
    V, H, W = 3, 128, 192
    _, proj, dv, _ = synth.make_inputs(V, H, W, seed=4, device=dev)
    imgs = torch.stack([synth.render_features(synth.make_scene(V, H, W, s), 1, 3, noise=0.02, device=dev, dtype=torch.float32)[0] for s in (4, 9)])
    proj2 = {k: v.expand(2, *v.shape[1:]).contiguous() for k, v in proj.items()}
    both = net(imgs, proj2, dv.expand(2, -1).contiguous(), tmp=[5.0, 5.0, 5.0, 1.0])["refined_depth"]
    for b in range(2):
        one = net(imgs[b:b + 1], proj, dv, tmp=[5.0, 5.0, 5.0, 1.0])["refined_depth"]
        assert torch.equal(both[b:b + 1], one), b
