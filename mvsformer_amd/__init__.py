"""mvsformer_amd — MI355X-native (gfx950) plane-sweep cost-volume path of MVSFormer.

Hand-written HIP kernels behind a C ABI (``include/mvs_hip.h`` -> ``libmvs_hip.so``) re-exposed with the
reference's own Python interface: ``StageNet`` (alias ``DepthNet``), ``CostRegNet``, ``CostRegNet3D``,
``homo_warping_3D_with_mask`` (alias ``homo_warping``), ``depth_regression``, the inverse-depth schedulers, and
``install()`` to rebind them inside an unmodified reference checkout.  ``synth`` (pure torch) builds the
synthetic DTU-shaped inputs used by the tests and ``bench.py``.

Importing the package does not load the library; the first op call does and fails loudly if it is missing.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "StageNet", "DepthNet", "CostRegNet", "CostRegNet3D", "CascadeMVS", "homo_warping_3D_with_mask",
           "homo_warping_3D", "homo_warping", "depth_regression", "conf_regression", "init_inverse_range",
           "schedule_inverse_range", "install", "fusion", "FPNDecoder", "FPNDecoderV2", "FPNEncoder", "vit_small", "VisionTransformer",
           "VITDecoderStage4Single", "DINOMVSNet"]


def __getattr__(name):
    if name in ("StageNet", "DepthNet"):
        from . import stagenet
        return getattr(stagenet, name)
    if name in ("CostRegNet", "CostRegNet3D", "Conv3d", "Deconv3d", "ConvBnReLU", "depth_regression", "conf_regression",
                "init_inverse_range", "schedule_inverse_range"):
        from . import module
        return getattr(module, name)
    if name in ("homo_warping_3D_with_mask", "homo_warping_3D", "homo_warping", "diff_homo_warping_3D_with_mask"):
        from . import warping
        return getattr(warping, name)
    if name in ("CascadeMVS", "randomize_bn_"):
        from . import cascade
        return getattr(cascade, name)
    if name in ("FPNDecoder", "FPNDecoderV2", "FPNEncoder"):
        from . import fpn
        return getattr(fpn, name)
    if name in ("vit_small", "VisionTransformer", "VITDecoderStage4Single"):
        from . import vit
        return getattr(vit, name)
    if name == "DINOMVSNet":
        from . import mvsformer_model
        return mvsformer_model.DINOMVSNet
    if name == "fusion":
        import importlib
        return importlib.import_module(".fusion", __name__)
    if name == "install":
        from .install import install
        return install
    raise AttributeError(name)
