"""ORACLE — test infrastructure, not product code.

Deterministic StageNet weights for the golden vectors.  The golden files do
not store the ~291k regularizer parameters; they store a seed, and both
``oracle/gen_golden.py`` (feeding the real reference modules) and the tests
(feeding the oracle and the HIP modules) rebuild the same ``state_dict`` from
it here.  Key names and shapes come from ``tests/golden/state_dict_shapes.json``,
which ``gen_golden.py`` dumped from the reference's own modules
(``StageNet(...).state_dict()``), so a drift in either side's parameter
layout fails ``tests/test_state_dict.py``.

BatchNorm running statistics and affine parameters are randomized on purpose:
PyTorch's default BN init makes eval-mode BN the identity and would hide
epilogue bugs.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict

import torch

_SHAPES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "state_dict_shapes.json")


def load_shapes(kind: str) -> Dict[str, list]:
    """``kind`` in {'stage_costregnet', 'stage_costregnet3d'} -> ``{key: shape}`` in the reference's order."""
    with open(_SHAPES) as f:
        return json.load(f)[kind]


def make_state_dict(shapes: Dict[str, list], seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key in shapes:                      # json preserves the reference's registration order
        shp = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.zeros(shp, dtype=torch.int64)
        elif key.endswith("running_var"):
            sd[key] = 0.5 + torch.rand(shp, generator=g)
        elif key.endswith("running_mean"):
            sd[key] = 0.2 * torch.randn(shp, generator=g)
        elif len(shp) == 1 and key.endswith("weight"):
            sd[key] = 0.5 + torch.rand(shp, generator=g)            # only BN gammas are 1-D weights
        elif len(shp) == 1:
            sd[key] = 0.2 * torch.randn(shp, generator=g)           # conv biases, BN betas
        else:
            # conv / deconv kernels: He-style scale on the receptive field so activations stay O(1)
            if ".conv7" in key or ".conv9" in key or ".conv11" in key:
                fan = shp[0] * math.prod(shp[2:]) / 4.0              # transposed conv: ~27/4..27/8 taps per output
            else:
                fan = shp[1] * math.prod(shp[2:])
            sd[key] = torch.randn(shp, generator=g) * math.sqrt(2.0 / max(fan, 1.0))
    return sd


_VIT_SHAPES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vit_shapes.json")


def load_vit_shapes(kind: str) -> Dict[str, list]:
    """``kind`` in {'vit_small', 'vit_decoder'}: key -> shape of the reference modules' ``state_dict()`` (``vits.vit_small(patch_size=16)``,
    ``VITDecoderStage4Single``), dumped by ``oracle/gen_golden.py`` from the reference's own classes."""
    with open(_VIT_SHAPES) as f:
        return json.load(f)[kind]


def make_vit_state_dict(shapes: Dict[str, list], seed: int) -> Dict[str, torch.Tensor]:
    """Seeded weights for the DINO ViT-small branch (21.7 M + 3.4 M parameters: not stored, rebuilt from the seed on both sides).  Linear
    layers get unit-gain fan-in scaling (the default trunc_normal(0.02) init makes every attention map uniform and the test blind), the
    q/k/v projection a gain of 2 so that the softmax is peaked; LayerNorm / BatchNorm parameters and running statistics are randomized."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key in shapes:
        shp = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.zeros(shp, dtype=torch.int64)
        elif key.endswith("running_var"):
            sd[key] = 0.5 + torch.rand(shp, generator=g)
        elif key.endswith("running_mean"):
            sd[key] = 0.2 * torch.randn(shp, generator=g)
        elif key in ("cls_token", "pos_embed"):
            sd[key] = 0.5 * torch.randn(shp, generator=g)
        elif len(shp) == 1 and key.endswith("weight"):
            sd[key] = 0.5 + torch.rand(shp, generator=g)            # LayerNorm / BatchNorm gains
        elif len(shp) == 1:
            sd[key] = 0.1 * torch.randn(shp, generator=g)           # biases
        else:
            if key.startswith("decoder.") and len(shp) == 4:        # ConvTranspose2d k4 s2: 4 taps per output
                fan = shp[0] * 4.0
            else:
                fan = float(math.prod(shp[1:]))
            gain = 2.0 if key.endswith("attn.qkv.weight") else 1.0
            sd[key] = torch.randn(shp, generator=g) * (gain / math.sqrt(max(fan, 1.0)))
    return sd


_MODEL_SHAPES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "dinomvsnet_shapes.json")


def load_model_shapes() -> Dict[str, list]:
    """key -> shape of the reference's ``DINOMVSNet(configs/config_mvsformer-p.json).state_dict()`` (582 keys, 26.5 M parameters), dumped by
    ``oracle/gen_golden.py::gen_end_to_end`` from the real class."""
    with open(_MODEL_SHAPES) as f:
        return json.load(f)


def make_model_state_dict(shapes: Dict[str, list], seed: int) -> Dict[str, torch.Tensor]:
    """Seeded weights for the whole MVSFormer-P model: each sub-module's keys go through the generator its own goldens use (``vit.`` /
    ``decoder_vit.`` -> :func:`make_vit_state_dict`, everything else -> :func:`make_state_dict`), with a seed per prefix."""
    groups: Dict[str, Dict[str, list]] = {}
    for key, shp in shapes.items():                      # prefix = first component, "fusions.<i>" for the stage networks
        parts = key.split(".")
        prefix = ".".join(parts[:2]) if parts[0] == "fusions" else parts[0]
        groups.setdefault(prefix, {})[key[len(prefix) + 1:]] = shp
    sd = {}
    for n, (prefix, sub) in enumerate(groups.items()):
        if prefix == "decoder_vit":                      # make_vit_state_dict recognises the transposed convolutions by the name "decoder."
            part = make_vit_state_dict(sub, seed + n)
        elif prefix == "vit":
            part = make_vit_state_dict(sub, seed + n)
        else:
            part = make_state_dict(sub, seed + n)
        for k, v in part.items():
            sd[prefix + "." + k] = v
    return sd
