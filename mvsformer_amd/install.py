"""Drop the MI355X path into an unmodified reference checkout.

``models/mvsformer_model.py`` binds its hot-path symbols at import time (``from models.module import *`` and
``from models.warping import homo_warping_3D_with_mask``, reference lines 6-7) and constructs ``StageNet`` by
name (lines 203 / 347).  ``install()`` rebinds exactly those names in the already-imported reference modules, so
``DINOMVSNet`` / ``TwinMVSNet`` build their ``fusions`` from the HIP-backed classes and ``load_state_dict`` of a
reference checkpoint keeps working (identical keys).  See INTEGRATION.md.
"""
from __future__ import annotations

import importlib

from . import fpn as _f
from . import module as _m
from . import vit as _v
from . import stagenet as _s
from . import warping as _w

_SYMBOLS = {
    "StageNet": _s.StageNet,
    "CostRegNet": _m.CostRegNet,
    "CostRegNet3D": _m.CostRegNet3D,
    "Conv3d": _m.Conv3d,
    "Deconv3d": _m.Deconv3d,
    "depth_regression": _m.depth_regression,
    "conf_regression": _m.conf_regression,
    "init_inverse_range": _m.init_inverse_range,
    "schedule_inverse_range": _m.schedule_inverse_range,
    "homo_warping_3D_with_mask": _w.homo_warping_3D_with_mask,
}
# the rows before the path: rebound on request (eval and training mode are both built; the ViT itself is frozen / eval-only)
_FEATURE_SYMBOLS = {"FPNDecoder": _f.FPNDecoder, "FPNDecoderV2": _f.FPNDecoderV2, "FPNEncoder": _f.FPNEncoder,
                    "VITDecoderStage4Single": _v.VITDecoderStage4Single}
# DINOMVSNet builds its backbone as ``vits.__dict__[vit_arch](...)`` (mvsformer_model.py:180): the factory is rebound in that module
_VIT_MODULE, _VIT_FACTORIES = "models.vision_transformer", {"vit_small": _v.vit_small}


def install(model_module: str = "models.mvsformer_model", also=("models.module", "models.warping"), features: bool = False) -> dict:
    """Rebind the hot-path names inside the reference's modules.  Returns ``{module: [names rebound]}``.
    ``features=True`` also rebinds ``FPNEncoder`` / ``FPNDecoder`` / ``VITDecoderStage4Single``, the ``vit_small`` factory of
    ``models.vision_transformer`` (the DINO branch of MVSFormer-P) and ``DINOMVSNet`` itself.  The FPN and the ViT decoder run in eval AND
    training mode; the ViT is eval-only (MVSFormer-P freezes it, ``"fix": true``; a model that fine-tunes the ViT must keep the reference's)."""
    done = {}
    symbols = dict(_SYMBOLS, **(_FEATURE_SYMBOLS if features else {}))
    for name in (model_module,) + tuple(also):
        try:
            mod = importlib.import_module(name)
        except ImportError:
            continue
        hit = []
        for sym, obj in symbols.items():
            if hasattr(mod, sym):
                setattr(mod, sym, obj)
                hit.append(sym)
        for sym in ("homo_warping_3D", "diff_homo_warping_3D_with_mask"):
            if hasattr(mod, sym):
                setattr(mod, sym, getattr(_w, sym))
                hit.append(sym)
        done[name] = hit
    if features:
        try:                                                 # the whole model too: its forward uses the CLS-row attention path (forward_with_cls_att)
            from .mvsformer_model import DINOMVSNet
            mod = importlib.import_module(model_module)
            if hasattr(mod, "DINOMVSNet"):
                mod.DINOMVSNet = DINOMVSNet
                done.setdefault(model_module, []).append("DINOMVSNet")
        except ImportError:
            pass
        try:
            vmod = importlib.import_module(_VIT_MODULE)
            for sym, obj in _VIT_FACTORIES.items():
                setattr(vmod, sym, obj)
            done[_VIT_MODULE] = list(_VIT_FACTORIES)
        except ImportError:
            pass
    return done
