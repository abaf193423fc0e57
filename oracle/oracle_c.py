"""ctypes binding of oracle/libsc_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/sc_oracle.c.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsc_oracle.so")
_lib = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libsc_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.sco_mask_stage.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, u8p]
        L.sco_mask_stage.restype = C.c_int
        L.sco_build_rhs.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int,
                                    u8p, i32p, f32p, f32p]
        L.sco_build_rhs.restype = C.c_int
        L.sco_fold.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p]
        L.sco_fold.restype = None
        L.sco_solve_dst.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.sco_solve_dst.restype = None
        L.sco_jacobi.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.sco_jacobi.restype = None
        L.sco_rbgs.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        L.sco_rbgs.restype = None
        L.sco_residual.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, f64p]
        L.sco_residual.restype = None
        L.sco_finish.argtypes = [u8p, C.c_int, f32p, i32p]
        L.sco_finish.restype = None
        L.sco_seamless_clone.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int,
                                         u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.sco_seamless_clone.restype = C.c_int
        L.sco_seamless_clone2.argtypes = L.sco_seamless_clone.argtypes + [C.c_int]
        L.sco_seamless_clone2.restype = C.c_int
        L.sco_solve_dst2.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
        L.sco_solve_dst2.restype = None
        L.sco_solve_dst3.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int]
        L.sco_solve_dst3.restype = None
        L.sco_seamless_clone3.argtypes = L.sco_seamless_clone2.argtypes + [C.c_int]
        L.sco_seamless_clone3.restype = C.c_int
        L.sco_mask_stage2.argtypes = L.sco_mask_stage.argtypes + [C.c_int]
        L.sco_mask_stage2.restype = C.c_int
        L.sco_build_rhs2.argtypes = L.sco_build_rhs.argtypes + [C.c_int]
        L.sco_build_rhs2.restype = C.c_int
        L.sco_seamless_clone4.argtypes = L.sco_seamless_clone3.argtypes + [C.c_int]
        L.sco_seamless_clone4.restype = C.c_int
        L.sco_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(u8p)


def _f32(a):
    return a.ctypes.data_as(f32p)


def mask_stage(mask: np.ndarray, cx: int, cy: int, opencv_grey: bool = False):
    """opencv_grey: OpenCV's grey-mask semantics (7x7 minimum filter) instead of the reference's thresholding erodes."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    mh, mw = mask.shape[:2]
    geo = np.zeros(6, np.int32)
    M = np.zeros(mw * mh, np.uint8)
    rc = lib().sco_mask_stage2(_u8(mask), mw, mh, mask.strides[0], cx, cy, geo.ctypes.data_as(i32p), _u8(M), int(opencv_grey))
    if rc:
        raise ValueError(f"sco_mask_stage rc={rc}")
    W, H = int(geo[2]), int(geo[3])
    return geo, M[:W * H].reshape(H, W).copy()


def build_rhs(dst, patch, geo, M, opencv_grey=False):
    dst = np.ascontiguousarray(dst, np.uint8)
    patch = np.ascontiguousarray(patch, np.uint8)
    M = np.ascontiguousarray(M, np.uint8)
    W, H = int(geo[2]), int(geo[3])
    B = np.zeros((3, H, W), np.float32)
    lap = np.zeros((3, H, W), np.float32)
    rc = lib().sco_build_rhs2(_u8(dst), dst.shape[1], dst.shape[0], dst.strides[0],
                              _u8(patch), patch.shape[1], patch.shape[0], patch.strides[0],
                              _u8(M), geo.ctypes.data_as(i32p), _f32(B), _f32(lap), int(opencv_grey))
    if rc:
        raise ValueError(f"sco_build_rhs rc={rc}")
    return B, lap


def fold(B, lap):
    _, H, W = B.shape
    g = np.zeros((3, H - 2, W - 2), np.float32)
    lib().sco_fold(_f32(B), _f32(lap), W, H, _f32(g))
    return g


INTERNALS = {"f64": 0, "f32": 1, "f32_bluestein": 2}


def solve_dst(g, nthreads=1, exact_den=False, internals="f64"):
    """exact_den=False reproduces OpenCV's float32 eigenvalue tables (see sc_oracle.c).
    internals: "f64" = transforms in double; "f32" = float32 mixed-radix transforms (what OpenCV's dft / cuFFT run),
    "f32_bluestein" = float32 chirp-z transforms (cuFFT's route for lengths with large prime factors)."""
    g = np.ascontiguousarray(g, np.float32)
    Cc, h, w = g.shape
    u = np.zeros_like(g)
    lib().sco_solve_dst3(_f32(g), w, h, Cc, _f32(u), nthreads, int(exact_den), INTERNALS[internals])
    return u


def jacobi(U, lap, sweeps):
    U = np.array(U, np.float32, copy=True, order="C")
    lap = np.ascontiguousarray(lap, np.float32)
    Cc, H, W = U.shape
    lib().sco_jacobi(_f32(U), _f32(lap), W, H, Cc, sweeps)
    return U


def rbgs(U, lap, sweeps, omega=1.0):
    U = np.array(U, np.float32, copy=True, order="C")
    lap = np.ascontiguousarray(lap, np.float32)
    Cc, H, W = U.shape
    lib().sco_rbgs(_f32(U), _f32(lap), W, H, Cc, sweeps, omega)
    return U


def residual(U, lap):
    U = np.ascontiguousarray(U, np.float32)
    lap = np.ascontiguousarray(lap, np.float32)
    Cc, H, W = U.shape
    out = np.zeros(2, np.float64)
    lib().sco_residual(_f32(U), _f32(lap), W, H, Cc, out.ctypes.data_as(f64p))
    return float(out[0]), float(out[1])


def finish(dst, U, geo):
    """clamp -> truncate -> splice the interior of the planar field U (3,H,W) into dst IN PLACE (sco_finish)."""
    assert dst.dtype == np.uint8 and dst.flags.c_contiguous and dst.flags.writeable
    U = np.ascontiguousarray(U, np.float32)
    geo = np.ascontiguousarray(geo, np.int32)
    lib().sco_finish(_u8(dst), dst.strides[0], _f32(U), geo.ctypes.data_as(i32p))
    return dst


def seamless_clone(dst, patch, mask, cx, cy, nthreads=1, exact_den=False, internals="f64", opencv_grey=False):
    """Returns a new blended image (input dst is not modified).  internals: see solve_dst; opencv_grey: see mask_stage."""
    out = np.array(dst, np.uint8, copy=True, order="C")
    patch = np.ascontiguousarray(patch, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    rc = lib().sco_seamless_clone4(_u8(patch), patch.shape[1], patch.shape[0], patch.strides[0],
                                   _u8(out), out.shape[1], out.shape[0], out.strides[0],
                                   _u8(mask), mask.shape[1], mask.shape[0], mask.strides[0], cx, cy, nthreads,
                                   int(exact_den), INTERNALS[internals], int(opencv_grey))
    if rc:
        raise ValueError(f"sco_seamless_clone rc={rc}")
    return out


def max_threads() -> int:
    return int(lib().sco_max_threads())
