#!/usr/bin/env python3
"""Regenerates the oracle-derived golden vectors in this directory.

The reference ships no lossless output, so these vectors are produced by the float64
oracle (oracle/oracle_np.py) -- itself pinned against the reference's committed JPEG -- and frozen
here so that neither the oracle, the C restatement nor the HIP path can drift unnoticed.

    python tests/golden/make_golden.py        # rewrites c1_expected.npz, synthetic_cases.npz, float_table_case.npz, c1_float_tables.npz
    python tests/golden/make_golden.py c1_float_tables     # only the named file(s)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_np as o  # noqa: E402


def c1_float_tables(sky, air, mask):
    """Round 3: config 1 in the reference's OWN arithmetic (float32 eigenvalue tables, seamlessClone_imp.cpp:596-599, :1651-1653).
    c1_expected.npz holds the exact system's bytes; the two differ by one in 127 channel values of the 298 x 192 ROI, so the
    default path (which returns the reference's arithmetic) is held to this one."""
    out, info = o.seamless_clone(sky, air, mask, 800, 150, return_all=True, float_tables=True)
    g = info["geo"]
    ex = o.seamless_clone(sky, air, mask, 800, 150)
    roi = (slice(g["lty"], g["lty"] + g["H"]), slice(g["ltx"], g["ltx"] + g["W"]))
    np.savez_compressed(os.path.join(HERE, "c1_float_tables.npz"), roi_bgr=out[roi],
                        differs_from_exact=np.int64(np.abs(out[roi].astype(int) - ex[roi].astype(int)).sum()),
                        sha256=np.frombuffer(hashlib.sha256(out.tobytes()).digest(), np.uint8))


def main(only=()):
    from PIL import Image
    sky = np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "sky.jpg")).convert("RGB"))[:, :, ::-1])
    air = np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "airplane.jpg")).convert("RGB"))[:, :, ::-1])
    mask = np.full(air.shape[:2], 255, np.uint8)
    if not only or "c1_float_tables" in only:
        c1_float_tables(sky, air, mask)
        print("wrote c1_float_tables.npz")
    if only and set(only) <= {"c1_float_tables"}:
        return
    out, info = o.seamless_clone(sky, air, mask, 800, 150, return_all=True)
    g = info["geo"]
    roi = out[g["lty"]:g["lty"] + g["H"], g["ltx"]:g["ltx"] + g["W"]]
    twice = o.seamless_clone(out, air, mask, 800, 150)
    np.savez_compressed(os.path.join(HERE, "c1_expected.npz"),
                        geo=np.array([g[k] for k in ("x0", "y0", "W", "H", "ltx", "lty")], np.int32),
                        roi_bgr=roi, roi_twice_bgr=twice[g["lty"]:g["lty"] + g["H"], g["ltx"]:g["ltx"] + g["W"]],
                        rhs_g_f32=info["g"].astype(np.float32),
                        eroded_mask_sum=np.int64(info["geo"]["M"].astype(np.int64).sum()),
                        dst_md5=np.frombuffer(hashlib.md5(sky.tobytes()).digest(), np.uint8))
    cases = {}
    for name, (W, H, ell) in {"r16x12": (16, 12, False), "r33x17": (33, 17, False), "e40x37": (40, 37, True)}.items():
        dst, patch, m, cx, cy = o.synth_inputs(W, H, margin=24, ellipse=ell)
        res, inf = o.seamless_clone(dst, patch, m, cx, cy, return_all=True)
        cases[name + "_dst"] = dst; cases[name + "_patch"] = patch; cases[name + "_mask"] = m
        cases[name + "_center"] = np.array([cx, cy], np.int32)
        cases[name + "_out"] = res
        cases[name + "_lap_f32"] = inf["lap"].astype(np.float32)
        cases[name + "_eroded"] = inf["geo"]["M"]
    np.savez_compressed(os.path.join(HERE, "synthetic_cases.npz"), **cases)
    # The reference's own arithmetic (float32 eigenvalue tables, seamlessClone_imp.cpp:596-599, :1651-1653) at a size where it is
    # visibly NOT the exact system's: inputs are regenerated from their seeds (not stored); frozen are a 96 x 96 crop of the
    # float-table result, a digest of the whole image and how many channels the two answers differ in.
    W, H = 1024, 700
    dst, patch, m, cx, cy = o.synth_inputs(W, H, margin=48)
    ft = o.seamless_clone(dst, patch, m, cx, cy, float_tables=True)
    ex = o.seamless_clone(dst, patch, m, cx, cy)
    y0, x0 = dst.shape[0] // 2 - 48, dst.shape[1] // 2 - 48
    np.savez_compressed(os.path.join(HERE, "float_table_case.npz"),
                        size=np.array([W, H, 48], np.int32), crop_origin=np.array([y0, x0], np.int32),
                        crop_float_tables=ft[y0:y0 + 96, x0:x0 + 96], crop_exact=ex[y0:y0 + 96, x0:x0 + 96],
                        sha256_float_tables=np.frombuffer(hashlib.sha256(ft.tobytes()).digest(), np.uint8),
                        channels_differing=np.int64((ft != ex).sum()), maxdiff=np.int64(np.abs(ft.astype(int) - ex.astype(int)).max()))
    print("wrote c1_expected.npz, synthetic_cases.npz, float_table_case.npz")


if __name__ == "__main__":
    main(tuple(sys.argv[1:]))
