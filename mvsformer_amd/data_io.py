"""On-disk formats either side of the depth-map filter: PFM depth maps, ``*_cam.txt`` camera files,
``pair.txt`` view graphs and the per-scan folder layout ``test.py`` writes and its ``TTDataset`` reads back.

Host-side only (numpy); same function names, arguments and byte-level output as the reference:
``read_pfm`` / ``save_pfm`` (datasets/data_io.py:7-71), ``write_cam`` (test.py:149-167), ``read_camera_parameters``
(test.py:102-112), ``read_pair_file`` (test.py:136-146); ``load_filter_sample`` assembles what ``TTDataset.__getitem__``
(test.py:347-401) returns, minus the RGB image decode (PIL is not a dependency here).
"""
from __future__ import annotations

import os
import re
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np


def read_pfm(filename) -> Tuple[np.ndarray, float]:
    """-> ``(data [H,W] or [H,W,3] float32, top row first; scale)``.  Raises on a bad magic or header."""
    with open(filename, "rb") as f:
        magic = f.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().rstrip())
        order = "<" if scale < 0 else ">"                      # negative scale marks little-endian samples
        data = np.fromfile(f, order + "f")
    shape = (height, width, 3) if magic == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename, image: np.ndarray, scale: float = 1) -> None:
    """Rows are stored bottom-up; greyscale ``[H,W]`` / ``[H,W,1]`` -> ``Pf``, ``[H,W,3]`` -> ``PF``; float32 only."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = "PF"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = "Pf"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(("%s\n%d %d\n%f\n" % (magic, image.shape[1], image.shape[0], -scale if little else scale)).encode("utf-8"))
        np.flipud(image).tofile(f)


def write_cam(file, cam) -> None:
    """``cam [2,4,4]``: extrinsic block, intrinsic 3x3 block, then the depth-range row ``cam[1][3]``."""
    rows = ["extrinsic"]
    rows += ["".join(str(cam[0][i][j]) + " " for j in range(4)) for i in range(4)]
    rows += ["", "intrinsic"]
    rows += ["".join(str(cam[1][i][j]) + " " for j in range(3)) for i in range(3)]
    rows += ["", " ".join(str(cam[1][3][j]) for j in range(4))]
    with open(file, "w") as f:
        f.write("\n".join(rows) + "\n")


def read_camera_parameters(filename) -> Tuple[np.ndarray, np.ndarray]:
    """-> ``(intrinsics [3,3], extrinsics [4,4])`` float32 from lines 1-4 and 7-9 of a ``*_cam.txt``."""
    with open(filename) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics


def read_pair_file(filename) -> List[Tuple[int, List[int]]]:
    """``pair.txt``: count, then per view its id and ``n id score id score ...``; views without sources are dropped."""
    out = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref = int(f.readline().rstrip())
            srcs = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if srcs:
                out.append((ref, srcs))
    return out


def _cam_2x4x4(path) -> np.ndarray:
    K, E = read_camera_parameters(path)
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    cam[0] = E
    cam[1, :3, :3] = K
    cam[1, 3, 3] = 1.0
    return cam


def _confidence(scan_folder, vid) -> np.ndarray:
    path = os.path.join(scan_folder, "confidence/{:0>8}.npy".format(vid))
    if not os.path.exists(path):
        path = os.path.join(scan_folder, "confidence_v2/{:0>8}.npy".format(vid))
    return np.asarray(np.load(path), dtype=np.float32).transpose(2, 0, 1)          # [H,W,C] on disk -> [C,H,W]


def load_filter_sample(scan_folder, id_ref: int, id_srcs: Sequence[int], n_src_views: int = 10) -> Dict[str, np.ndarray]:
    """One ``TTDataset`` item (test.py:347-401) without the image: ``ref_depth [1,H,W]``, ``ref_cam [2,4,4]``,
    ``ref_conf [C,H,W]``, ``src_depths [V,1,H,W]``, ``src_cams [V,2,4,4]``, ``src_confs [V,C,H,W]``, ``ref_id``.
    Source views whose camera file is missing are skipped, as in the reference."""
    cam = lambda v: os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(v))  # noqa: E731
    depth = lambda v: np.array(read_pfm(os.path.join(scan_folder, "depth_est/{:0>8}.pfm".format(v)))[0], dtype=np.float32)  # noqa: E731
    srcs = [v for v in list(id_srcs)[:n_src_views] if os.path.exists(cam(v))]
    return {"ref_depth": depth(id_ref)[None], "ref_cam": _cam_2x4x4(cam(id_ref)), "ref_conf": _confidence(scan_folder, id_ref),
            "src_depths": np.stack([depth(v) for v in srcs])[:, None], "src_cams": np.stack([_cam_2x4x4(cam(v)) for v in srcs]),
            "src_confs": np.stack([_confidence(scan_folder, v) for v in srcs]), "ref_id": id_ref}


def save_depth_outputs(scan_folder, vid: int, depth: np.ndarray, confidences: np.ndarray, cam: np.ndarray) -> None:
    """What ``save_depth`` leaves per view for the filter to pick up (test.py:296-318): ``depth_est/%08d.pfm``,
    ``confidence/%08d.npy`` ([H,W,C]) and ``cams/%08d_cam.txt``."""
    for sub in ("depth_est", "confidence", "cams"):
        os.makedirs(os.path.join(scan_folder, sub), exist_ok=True)
    save_pfm(os.path.join(scan_folder, "depth_est/{:0>8}.pfm".format(vid)), np.ascontiguousarray(depth, dtype=np.float32))
    np.save(os.path.join(scan_folder, "confidence/{:0>8}.npy".format(vid)), np.asarray(confidences, dtype=np.float32))
    write_cam(os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(vid)), cam)
