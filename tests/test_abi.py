"""CPU-side checks of the C-ABI boundary: the header, the ctypes table and the built library agree, the
package mirrors the reference's interface (names, signatures, state_dict keys), and there is no CPU fallback."""
import inspect
import json
import os
import re

import pytest
import torch

from conftest import GOLDEN, REPO

HEADER = os.path.join(REPO, "include", "mvs_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|int64_t|char\s*\*|const char\s*\*)\s+(mvs_\w+)\s*\(", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"(mvs_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return names, protos


def test_header_matches_ctypes_table():
    from mvsformer_amd import _lib
    names, protos = header_functions()
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    for n, (_, args) in _lib.SIGNATURES.items():
        assert protos[n] == len(args), (n, protos[n], len(args))


def test_library_exports_every_symbol():
    from mvsformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()               # binds every symbol of SIGNATURES or raises
    assert lib.mvs_version() == _lib.ABI_VERSION
    names, _ = header_functions()
    for n in names:
        assert hasattr(lib, n)
    assert lib.mvs_conv3d_packed_floats(8, 16, 0) == 2 * 27 * 4 * 16
    assert lib.mvs_conv3d_packed_floats(64, 64, 0) == 16 * 27 * 4 * 80
    assert lib.mvs_conv3d_packed_floats(16, 8, 2) == 4 * 3 * 4 * 80    # 2 chunks of 8 channels = 4 slabs


def test_state_dict_keys_match_reference():
    import mvsformer_amd as m
    shapes = json.load(open(os.path.join(GOLDEN, "state_dict_shapes.json")))
    args = dict(base_ch=8, fusion_type="cnn", depth_type="ce")
    for kind, nd in (("stage_costregnet", 16), ("stage_costregnet3d", 4)):
        net = m.StageNet(dict(args), nd, 0)
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert list(mine.keys()) == list(shapes[kind].keys())
        assert mine == shapes[kind]
    assert m.DepthNet is m.StageNet and m.homo_warping is m.homo_warping_3D


def test_reference_signatures():
    import mvsformer_amd as m
    assert list(inspect.signature(m.StageNet.__init__).parameters) == ["self", "args", "ndepth", "stage_idx"]
    assert list(inspect.signature(m.StageNet.forward).parameters) == ["self", "features", "proj_matrices", "depth_values", "tmp"]
    assert inspect.signature(m.StageNet.forward).parameters["tmp"].default == 2.0
    assert list(inspect.signature(m.homo_warping_3D_with_mask).parameters) == ["src_fea", "src_proj", "ref_proj", "depth_values"]
    assert list(inspect.signature(m.CostRegNet.__init__).parameters) == ["self", "in_channels", "base_channels", "last_layer"]
    assert list(inspect.signature(m.CostRegNet3D.__init__).parameters) == ["self", "in_channels", "base_channel"]
    assert list(inspect.signature(m.depth_regression).parameters) == ["p", "depth_values"]
    assert list(inspect.signature(m.schedule_inverse_range).parameters) == ["depth", "depth_hypo", "ndepths", "split_itv", "H", "W"]
    with pytest.raises(NotImplementedError):
        m.StageNet(dict(base_ch=8, fusion_type="epipole", depth_type="ce"), 8, 0)


def test_no_cpu_fallback():
    """CPU tensors must be refused loudly, not silently computed some other way."""
    import mvsformer_amd as m
    from mvsformer_amd._lib import MvsHipError
    with pytest.raises(MvsHipError):
        m.homo_warping_3D_with_mask(torch.zeros(1, 8, 4, 4), torch.eye(4)[None], torch.eye(4)[None], torch.ones(1, 2))
    net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), 4, 0).eval()
    with pytest.raises(MvsHipError):
        net(torch.zeros(1, 3, 8, 8, 8), torch.zeros(1, 3, 2, 4, 4), torch.ones(1, 4, 8, 8))
    net.train()
    with pytest.raises(MvsHipError):
        net(torch.zeros(1, 3, 8, 8, 8), torch.zeros(1, 3, 2, 4, 4), torch.ones(1, 4, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "mvsformer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout only exists in the build container")
def test_install_rebinds_reference_symbols():
    """install() swaps the hot-path names inside the unmodified reference modules (import side effects of unrelated
    files stubbed exactly as oracle/gen_golden.py does)."""
    import subprocess
    import sys
    code = r'''
import sys, types, torch.nn as nn
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(0, %r)
def stub(name, **kw):
    m = types.ModuleType(name); m.__dict__.update(kw); sys.modules[name] = m
stub("timm"); stub("timm.models")
stub("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x), trunc_normal_=nn.init.trunc_normal_)
stub("timm.models.vision_transformer", Block=nn.Module)
stub("torchvision"); stub("torchvision.utils"); stub("omegaconf", OmegaConf=object)
import models.mvsformer_model as mm
import mvsformer_amd
done = mvsformer_amd.install()
assert mm.StageNet is mvsformer_amd.StageNet and mm.homo_warping_3D_with_mask is mvsformer_amd.homo_warping_3D_with_mask
assert mm.CostRegNet is mvsformer_amd.CostRegNet and mm.CostRegNet3D is mvsformer_amd.CostRegNet3D
net = mm.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), 4, 0)
assert type(net.cost_reg).__module__.startswith("mvsformer_amd")
print("OK", sorted(done))
''' % REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_kernel_timer_per_step_medians():
    """bench.py's per-kernel figures: the j-th launch of a kernel name within a step is the median over the profile steps (a launch whose
    event pair also caught a pre-empted host thread must not move the figure), launches of different shapes under one name stay apart."""
    from mvsformer_amd.ops import KernelTimer
    # 5 steps x 3 launches of one name (three shapes: 0.03, 0.09, 0.14 ms); step 2's first launch waited 12 ms for the host
    ms = []
    for step in range(5):
        ms += [0.03 + 0.001 * step, 0.09, 0.14 - 0.001 * step]
    ms[2 * 3] = 12.0
    med = KernelTimer.per_step_medians(ms, 5)
    assert med is not None and len(med) == 3
    assert abs(med[0] - 0.033) < 1e-9 and med[1] == 0.09 and abs(med[2] - 0.138) < 1e-9
    assert abs(sum(ms) / 5 - sum(med)) > 2.0                      # the mean over the steps would have reported 2.65 ms for 0.26 ms of work
    assert KernelTimer.per_step_medians(ms[:-1], 5) is None        # launches do not divide into equal steps: caller keeps the plain mean
    assert KernelTimer.per_step_medians(ms[:6], 2) is None         # fewer than three steps: no median


def test_store_plan_and_vis_mode_host_logic(monkeypatch):
    """Host-side decisions of StageNet that need no GPU: which stages keep their per-view correlation volumes (config-2 stage 1: 127 MB,
    stored; stage 2: 254 MB, recomputed; the MVS_CV_STORE_MAX_MB override), and the MVS_VIS switch rejecting unknown values."""
    import pytest
    import mvsformer_amd as m
    from mvsformer_amd import stagenet
    monkeypatch.delenv("MVS_CV_STORE_MAX_MB", raising=False)
    f1 = torch.empty(1, 5, 144, 192, 64)                  # [B,V,H,W,C] channel-last, stage 1 of config 2
    f2 = torch.empty(1, 5, 288, 384, 32)
    f3 = torch.empty(1, 5, 576, 768, 16)
    monkeypatch.delenv("MVS_CV_STORE_BANDS", raising=False)
    assert stagenet._store_plan(f1, 32, 8) == 1                # one store for the whole image
    assert stagenet._store_plan(f2, 16, 8) == 0                # 254 MB: recompute (banding is opt-in)
    assert stagenet._store_plan(f3, 8, 8) == 0                 # C = 16: not built
    monkeypatch.setenv("MVS_CV_STORE_BANDS", "4")
    assert stagenet._store_plan(f2, 16, 8) == 2                # two bands of 144 + 7 rows: 133 MB each
    assert stagenet._store_plan(f1, 32, 8) == 1
    monkeypatch.setenv("MVS_CV_STORE_MAX_MB", "70")
    assert stagenet._store_plan(f2, 16, 8) == 4                # 72 + 7 rows: 69.7 MB
    assert stagenet._store_plan(f1, 32, 8) == 2
    monkeypatch.setenv("MVS_CV_STORE_MAX_MB", "400")
    assert stagenet._store_plan(f2, 16, 8) == 1
    monkeypatch.setenv("MVS_CV_STORE_MAX_MB", "0")
    assert stagenet._store_plan(f1, 32, 8) == 0
    # a small map with many hypotheses: the store would fit, but mvs_cv_corr_fwd's per-pixel LDS rows do not: the size query
    # says "not built" and the stage falls back to the recomputing sweeps instead of raising from the launch
    monkeypatch.setenv("MVS_CV_STORE_MAX_MB", "160")
    tiny = torch.empty(1, 3, 16, 24, 32)
    assert stagenet._store_plan(tiny, 256, 8) == 1
    assert stagenet._store_plan(tiny, 352, 8) == 0
    assert stagenet._store_plan(torch.empty(1, 3, 16, 24, 64), 1000, 8) == 0
    net = m.StageNet(dict(base_ch=8, fusion_type="cnn", depth_type="ce"), 8, 0).eval()
    monkeypatch.setenv("MVS_VIS", "winograd")
    with pytest.raises(ValueError):
        net._vis_params()


def test_small_limit_env_is_parsed_once_and_validated():
    """MVS_CONV_SMALL_MAX_WORK accepts 'conv,deconv' or one value for both, and rejects anything else at parse time."""
    from mvsformer_amd import module
    assert module._parse_small_limit(None) == module.SMALL_MAX_WORK
    assert module._parse_small_limit("0") == (0, 0)
    assert module._parse_small_limit("5,7") == (5, 7)
    for bad in ("a", "1,2,3", "-1", "1,x"):
        with pytest.raises(ValueError):
            module._parse_small_limit(bad)


def test_pack_table_host_fill_without_a_gpu():
    """The job table of the one-launch weight packing is filled on the HOST (mvs_bf16_pack_table_fill): block starts accumulate, a weight
    that does not fit the requested map is refused, 2-D kernels take 9 taps."""
    import ctypes
    from mvsformer_amd import _lib
    lib = _lib.load()
    n = 3
    nbytes = lib.mvs_bf16_pack_table_bytes(n)
    assert nbytes > 0 and lib.mvs_bf16_pack_table_bytes(0) < 0
    host = (ctypes.c_uint8 * nbytes)()
    fake = 0x1000                                            # never dereferenced on the host
    assert lib.mvs_bf16_pack_table_fill(host, n, 0, fake, 16, 8, 0, 16, 8, 27, fake) == 0       # conv 8 -> 16
    assert lib.mvs_bf16_pack_table_fill(host, n, 1, fake, 16, 1, 0, 16, 8, 9, fake) == 0        # 2-D 1 -> 16 run as 8 -> 16
    assert lib.mvs_bf16_pack_table_fill(host, n, 2, fake, 1, 8, 0, 8, 8, 27, fake) == 0         # 8 -> 1 run as 8 -> 8
    starts = (ctypes.c_int32 * (n + 1)).from_buffer(host, nbytes - 4 * (n + 1))
    want = [0]
    for cin, cout, taps in ((8, 16, 27), (8, 16, 9), (8, 8, 27)):
        want.append(want[-1] + (lib.mvs_bf16_packed_elems_taps(cin, cout, taps) // 8 + 255) // 256)
    assert list(starts) == want
    assert lib.mvs_bf16_pack_table_fill(host, n, 0, fake, 32, 8, 0, 16, 8, 27, fake) != 0       # 32 rows do not fit a 16-row map
    assert lib.mvs_bf16_pack_table_fill(host, n, 0, fake, 16, 8, 0, 16, 8, 5, fake) != 0        # taps must be 27 or 9


def test_struct_layouts_match_the_header(tmp_path):
    """The by-value job structs of the C ABI (``MvsWgradJob``, ``MvsAdamTensor``) as the C compiler lays them out against the ctypes twins
    in mvsformer_amd/_lib.py: sizes and every field offset (the header is plain C: gcc compiles it as such)."""
    import ctypes
    import shutil
    import subprocess
    from mvsformer_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    structs = {"MvsWgradJob": _lib.WgradJob, "MvsAdamTensor": _lib.AdamTensor}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mvs_hip.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f, _ in ct._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f))
        lines.append('printf("\\n");')
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-std=c99", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    for line in out:
        if not line.strip():
            continue
        parts = line.split()
        ct = structs[parts[0]]
        want = [int(v) for v in parts[1:]]
        got = [ctypes.sizeof(ct)] + [getattr(ct, f).offset for f, _ in ct._fields_]
        assert got == want, (parts[0], got, want)


def test_fused_adamw_and_queues_have_no_cpu_path():
    """Host-side contracts that need no GPU: FusedAdamW refuses CPU parameters (no fallback), amsgrad is refused at construction; an armed
    SkipLink that is never filled is an error by construction (checked in LayerBf16Fn.backward), and an empty WgradQueue flushes to nothing."""
    from mvsformer_amd import autograd as ag
    from mvsformer_amd._lib import MvsHipError
    from mvsformer_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.ones(3))
    with pytest.raises(MvsHipError):
        FusedAdamW([p], amsgrad=True)
    opt = FusedAdamW([p], lr=1e-3)
    p.grad = torch.ones(3)
    with pytest.raises(MvsHipError):
        opt.step()
    opt.zero_grad(set_to_none=True)
    opt.step()                                                  # nothing to do: no gradient, no launch
    link = ag.SkipLink()
    assert link.armed is False and link.grad is None
    q = ag.WgradQueue()
    q.flush(())                                                 # no jobs: no call into the library
    assert q.jobs == []


def test_fpn_split_form_shape_queries_and_argument_checks():
    """The FPN split-form entry points' pure host logic (no GPU): which layer shapes each kernel family is built for, the operand sizes, and
    that a bad call is refused with MVS_EINVAL and a message instead of launching anything."""
    from mvsformer_amd import _lib
    lib = _lib.load()
    enc_full = [(3, 8, 7, 1), (8, 8, 5, 1)]
    enc_below = [(16, 16, 3, 1), (32, 32, 3, 1), (64, 64, 3, 1), (8, 16, 5, 2), (16, 32, 5, 2), (32, 64, 3, 2)]
    for cfg in enc_full + enc_below:                          # every FPNEncoder layer shape has exactly one split-form kernel
        assert bool(lib.mvs_conv2d_x3_supported(*cfg)) == (cfg in enc_full), cfg
        assert bool(lib.mvs_conv2d_x3s_supported(*cfg)) == (cfg in enc_below), cfg
    assert not lib.mvs_conv2d_x3s_supported(16, 16, 3, 2) and not lib.mvs_conv2d_x3_supported(8, 8, 5, 2)
    assert lib.mvs_conv2d_x3_prepared_bytes(3, 8, 7) == lib.mvs_conv2d_x3_prepared_bytes(8, 8, 5) == 8 * 3 * 64 * 16
    assert lib.mvs_conv2d_x3s_prepared_bytes(64, 64, 3, 1) == 4 * 5 * 4 * 3 * 64 * 16 and lib.mvs_conv2d_x3s_prepared_bytes(8, 16, 5, 2) == 1 * 7 * 1 * 3 * 64 * 16
    assert lib.mvs_conv2d_x3s_prepared_bytes(8, 16, 5, 1) == -1
    assert lib.mvs_fpn_level_x3_prepared_bytes(8) == 4 * 7 * 3 * 64 * 16 and lib.mvs_fpn_level_x3_prepared_bytes(16) == -1
    assert lib.mvs_fpn_level_cp_prepared_bytes(8) == 39 * 64 * 16 and lib.mvs_fpn_level_cp_prepared_bytes(32) == -1
    assert lib.mvs_fpn_level_x3s_prepared_bytes(16) == 4 * 5 * 1 * 3 * 64 * 16 and lib.mvs_fpn_level_x3s_prepared_bytes(32) == 2 * lib.mvs_fpn_level_x3s_prepared_bytes(16)
    assert lib.mvs_fpn_level_x3s_prepared_bytes(8) == -1
    # null pointers / unsupported shapes: refused before any launch (MVS_EINVAL = a negative code, the message names the entry point)
    assert lib.mvs_fpn_level_cp(None, None, None, None, None, 1, 8, 4, 4, None, None) < 0 and b"mvs_fpn_level_cp" in lib.mvs_last_error()
    assert lib.mvs_conv2d_x3s_bn_lrelu(None, 0, None, None, 1, 16, 16, 3, 1, 8, 8, 0.1, None, None) < 0 and b"mvs_conv2d_x3s_bn_lrelu" in lib.mvs_last_error()
    assert lib.mvs_fpn_level_x3s(None, None, None, None, None, None, 1, 16, 4, 4, None, 0, None, None) < 0 and b"mvs_fpn_level_x3s" in lib.mvs_last_error()


def test_counter_traffic_evidence_matches_the_kernel_sources():
    """profiles/traffic_by_kernel.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/collect_profiles.sh) records a digest per source
    file; ``bench.py`` reports a kernel's HBM traffic only while the files that kernel is built from are unchanged (the driver's
    line carried ``traffic: null`` because a late edit had voided the evidence).  This test fails when an eval-path kernel source was edited
    after the last collection: re-run ``tools/collect_profiles.sh`` as the round's last GPU call and commit its output."""
    import json
    from mvsformer_amd import _sources
    path = os.path.join(REPO, "profiles", "traffic_by_kernel.json")
    tj = json.load(open(path))
    rec = tj["source_digests"]
    assert tj.get("collected_at_head")
    now = _sources.file_digests()
    stale = sorted(k for k in tj["kernels"] if not _sources.current(k, rec, now))
    assert not stale, "counter traffic is stale for %s: re-collect (tools/collect_profiles.sh)" % stale[:5]
    for k in ("vis_x3_kernel", "cv_aggregate_kernel<2,true>", "cv_entropy_kernel<2>", "nchw_to_nhwc_multi"):      # the bench line's roofline kernels
        assert k in tj["kernels"], k
        assert all(os.path.exists(os.path.join(_sources.CSRC, f)) for f in _sources.files_of(k))
    assert len(_sources.files_of("vis_x3_kernel")) < 6          # a known kernel maps to ITS files, not to all of csrc/
