#!/usr/bin/env python3
"""Regenerates the oracle-derived golden vectors in this directory.

The reference ships no lossless output, so these vectors are produced by the float64
oracle (oracle/oracle_np.py) -- itself pinned against the reference's committed JPEG -- and frozen
here so that neither the oracle, the C restatement nor the HIP path can drift unnoticed.

    python tests/golden/make_golden.py        # rewrites c1_expected.npz and synthetic_cases.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_np as o  # noqa: E402


def main():
    from PIL import Image
    sky = np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "sky.jpg")).convert("RGB"))[:, :, ::-1])
    air = np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "airplane.jpg")).convert("RGB"))[:, :, ::-1])
    mask = np.full(air.shape[:2], 255, np.uint8)
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
    print("wrote c1_expected.npz, synthetic_cases.npz")


if __name__ == "__main__":
    main()
