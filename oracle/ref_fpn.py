"""TEST INFRASTRUCTURE — CPU oracle for the FPN decoder and encoder (the steps that hand features to the path).

Restates reference models/module.py:273-302 (``FPNDecoderV2``, pinned by tests/golden/fpn_decoder_v2.npz) and models/module.py:242-270 (``FPNDecoder``: lateral 1x1 convolutions, bilinear x2 upsampling with
align_corners=True, 3x3 output convolutions followed by BatchNorm2d and Swish, module.py:200-206) as a function of the
module's ``state_dict``, eval-mode BatchNorm, and models/module.py:208-240 (``FPNEncoder``) likewise.  Pinned by
tests/golden/fpn_decoder.npz and fpn_encoder.npz (outputs of the real modules, made by oracle/gen_golden.py).  Never imported by the product path.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5   # nn.BatchNorm2d default, module.py:246-255 constructs it without arguments


def swish(x):
    """module.py:205-206"""
    return x * torch.sigmoid(x)


def _bn_eval(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
                        False, 0.0, BN_EPS)


def _out(x, sd, name, padding):
    """nn.Sequential(Conv2d, BatchNorm2d, Swish) — module.py:246,249,252,255"""
    y = F.conv2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=padding)
    return swish(_bn_eval(y, sd, name + ".1"))


def fpn_decoder_forward(sd, conv01, conv11, conv21, conv31):
    """module.py:257-270 -> [out0 (1/8), out1 (1/4), out2 (1/2), out3 (full)], all [N,C,H,W]."""
    intra = conv31
    out0 = _out(intra, sd, "out0", 0)
    outs = [out0]
    for k, lateral in ((1, conv21), (2, conv11), (3, conv01)):
        intra = F.interpolate(intra, scale_factor=2, mode="bilinear", align_corners=True) + \
            F.conv2d(lateral, sd["inner%d.weight" % k], sd["inner%d.bias" % k])
        outs.append(_out(intra, sd, "out%d" % k, 1))
    return outs


def fpn_decoder_v2_forward(sd, conv01, conv11, conv21, conv31, vit1, vit2, vit3):
    """module.py:291-302 (``FPNDecoderV2.forward``) -> [out1 (1/8), out2 (1/4), out3 (1/2), out4 (full)]; eval-mode BatchNorm.
    out_k = Swish(BN(conv3x3(.))) (module.py:276,280,284,288), upsample_k = ReLU(BN(ConvTranspose2d(4, 2, 1)(.))) (module.py:277-286)."""
    def up(x, name):
        y = F.conv_transpose2d(x, sd[name + ".0.weight"], sd[name + ".0.bias"], stride=2, padding=1)
        return F.relu(_bn_eval(y, sd, name + ".1"))

    out1 = _out(torch.cat([conv31, vit1], dim=1), sd, "out1", 1)
    out2 = _out(torch.cat([up(out1, "upsample1") + conv21, vit2], dim=1), sd, "out2", 1)
    out3 = _out(torch.cat([up(out2, "upsample2") + conv11, vit3], dim=1), sd, "out3", 1)
    out4 = _out(up(out3, "upsample3") + conv01, sd, "out4", 1)
    return [out1, out2, out3, out4]


def make_case(seed, N, h, w, feat_chs=(8, 16, 32, 64)):
    """Seeded encoder outputs for a decoder whose coarsest level is h x w: (conv01, conv11, conv21, conv31)."""
    g = torch.Generator().manual_seed(seed)
    return tuple(torch.randn(N, feat_chs[i], h * 2 ** (3 - i), w * 2 ** (3 - i), generator=g) for i in range(4))


def randomize_bn(module, seed):
    """Give every BatchNorm2d non-trivial affine parameters and running statistics (a fresh module has 1/0/0/1)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.num_features, generator=g)
            m.bias.data = 0.2 * torch.randn(m.num_features, generator=g)
            m.running_mean = 0.3 * torch.randn(m.num_features, generator=g)
            m.running_var = 0.5 + torch.rand(m.num_features, generator=g)


# --------------------------------------------------------------------------------------------- FPNEncoder (module.py:208-240)
ENCODER_LAYERS = (  # name, cin, cout, kernel, stride  (module.py:211-224; padding = kernel // 2 everywhere)
    ("conv00", 3, 8, 7, 1), ("conv01", 8, 8, 5, 1),
    ("downsample1", 8, 16, 5, 2), ("conv10", 16, 16, 3, 1), ("conv11", 16, 16, 3, 1),
    ("downsample2", 16, 32, 5, 2), ("conv20", 32, 32, 3, 1), ("conv21", 32, 32, 3, 1),
    ("downsample3", 32, 64, 3, 2), ("conv30", 64, 64, 3, 1), ("conv31", 64, 64, 3, 1))


def conv_bn_lrelu(x, sd, name, stride, padding):
    """models/module.py:40-73 ``Conv2d`` with norm_type='BN': conv (no bias) -> BatchNorm2d (eval) -> leaky_relu(0.1)."""
    y = F.conv2d(x, sd[name + ".conv.weight"], None, stride=stride, padding=padding)
    y = F.batch_norm(y, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                     False, 0.0, BN_EPS)
    return F.leaky_relu(y, 0.1)


def fpn_encoder_forward(sd, x):
    """module.py:226-240 -> [conv01 (full), conv11 (1/2), conv21 (1/4), conv31 (1/8)]."""
    feats = {}
    for name, _, _, k, s in ENCODER_LAYERS:
        x = conv_bn_lrelu(x, sd, name, s, k // 2)
        feats[name] = x
    return [feats["conv01"], feats["conv11"], feats["conv21"], feats["conv31"]]
