"""CPU: the numpy walk-throughs of the FPN kernels' index arithmetic (tools/sim_fpn.py, tools/sim_conv2d.py) stay in step with the oracle.

They restate, formula by formula, what csrc/fpn.hip and csrc/conv2d.hip compute per block / wavefront / lane (weight packing incl. the
row-pair N form, halo staging, coarse-window origin and bilinear tap offsets, stride-2 fragment addresses, epilogue lane mapping);
tools/sim_x3.py does the same for the regularizer's three-term bf16 split form (csrc/conv3d_x3.hip).  They are not tests of the HIP code (tests/test_hip_fpn.py is) - they pin the DESIGN of
those kernels on a machine without a GPU, which is how both kernels were parity-green on their first GPU run.
"""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fpn_decoder_walkthrough_matches_oracle(capsys):
    _load("sim_fpn").main()                      # asserts <= 1e-4 abs per level
    assert capsys.readouterr().out.strip().endswith("ok")


def test_fpn_encoder_conv_walkthrough_matches_torch(capsys):
    _load("sim_conv2d").main()                   # all eight layer shapes, odd sizes, <= 1e-5 abs
    assert capsys.readouterr().out.strip().endswith("ok")


def test_x3_split_form_walkthrough_matches_fp64(capsys):
    _load("sim_x3").main()                       # split exactness, packed-weight layout, K-block order, stride-2 column order; error vs fp64
    assert capsys.readouterr().out.strip().endswith("ok")
