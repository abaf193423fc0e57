"""OpenCV FileStorage YAML and 24-bit BMP I/O without OpenCV.

The reference exchanges inputs as cv::FileStorage yml so both sides see byte-identical pixels
(README.md:59, seamlessClone-OpenCV/jpg2yaml.cpp:12-22; reader takes node "data",
seamlessClone_imp.cu:226-237) and writes results as bottom-up 24-bit BMP
(seamlessClone_imp.cu:68-190).  PyYAML rejects the `%YAML:1.0` directive and the
`!!opencv-matrix` tag, hence the small regex reader.
"""
from __future__ import annotations

import gzip
import re
import struct

import numpy as np

_DT = {"u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16, "i": np.int32, "f": np.float32, "d": np.float64}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def read_yml(path, node: str = "data") -> np.ndarray:
    with _open(path, "rt") as f:
        txt = f.read()
    m = re.search(r"^\s*%s:\s*!!opencv-matrix\s*\n(.*?)(?=^\S|\Z)" % re.escape(node), txt, re.S | re.M)
    if not m:
        raise ValueError(f"{path}: no opencv-matrix node '{node}'")
    body = m.group(1)
    rows = int(re.search(r"rows:\s*(\d+)", body).group(1))
    cols = int(re.search(r"cols:\s*(\d+)", body).group(1))
    dt = re.search(r'dt:\s*"?(\d*)([ucwsifd])"?', body)
    ch = int(dt.group(1)) if dt.group(1) else 1
    dtype = _DT[dt.group(2)]
    data = re.search(r"data:\s*\[(.*?)\]", body, re.S).group(1)
    arr = np.array(data.replace("\n", " ").split(","), dtype=np.float64).astype(dtype)
    if arr.size != rows * cols * ch:
        raise ValueError(f"{path}: expected {rows * cols * ch} values, found {arr.size}")
    return arr.reshape(rows, cols, ch) if ch > 1 else arr.reshape(rows, cols)


def write_yml(path, mat: np.ndarray, node: str = "data", name: str | None = None) -> None:
    a = np.asarray(mat)
    ch = 1 if a.ndim == 2 else a.shape[2]
    code = {v: k for k, v in _DT.items()}[a.dtype.type]
    dt = f"{ch}{code}" if ch > 1 else code
    flat = a.reshape(-1)
    if a.dtype.kind == "f":
        vals = [repr(float(v)) for v in flat]
    else:
        vals = [str(int(v)) for v in flat]
    lines, cur = [], "   data: [ "
    for i, v in enumerate(vals):
        tok = v + (", " if i + 1 < len(vals) else " ]")
        if len(cur) + len(tok) > 78:
            lines.append(cur.rstrip())
            cur = "       "
        cur += tok
    lines.append(cur)
    with _open(path, "wt") as f:
        f.write("%YAML:1.0\n---\n")
        if name:
            f.write(f"mat_name: {name}\n")
        f.write(f"{node}: !!opencv-matrix\n   rows: {a.shape[0]}\n   cols: {a.shape[1]}\n   dt: \"{dt}\"\n" if ch > 1
                else f"{node}: !!opencv-matrix\n   rows: {a.shape[0]}\n   cols: {a.shape[1]}\n   dt: {dt}\n")
        f.write("\n".join(lines) + "\n")


def write_bmp(path, bgr: np.ndarray) -> None:
    """24-bit bottom-up BMP, BGR, rows padded to 4 bytes (seamlessClone_imp.cu:68-190)."""
    a = np.ascontiguousarray(bgr, np.uint8)
    h, w = a.shape[:2]
    pad = (-3 * w) % 4
    row = 3 * w + pad
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + row * h, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, row * h, 2835, 2835, 0, 0))
        for y in range(h - 1, -1, -1):
            f.write(a[y].tobytes() + b"\0" * pad)


def read_bmp(path) -> np.ndarray:
    with open(path, "rb") as f:
        d = f.read()
    if d[:2] != b"BM":
        raise ValueError("not a BMP")
    off = struct.unpack_from("<I", d, 10)[0]
    w, h = struct.unpack_from("<ii", d, 18)
    bpp = struct.unpack_from("<H", d, 28)[0]
    if bpp != 24:
        raise ValueError("only 24-bit BMP")
    row = (3 * w + 3) // 4 * 4
    flip = h > 0
    h = abs(h)
    img = np.frombuffer(d, np.uint8, row * h, off).reshape(h, row)[:, :3 * w].reshape(h, w, 3)
    return np.ascontiguousarray(img[::-1] if flip else img)
