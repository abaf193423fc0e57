"""Command line with the reference's argv (seamlessClone-CUDA/seamlessClone_main.cu:69-94):

    python -m seamlesscloneoptimization_amd.cli src.yml dst.yml mask.yml centerX centerY gpu [--out result.bmp]

src = patch ("face"), dst = destination ("body"), mask: OpenCV FileStorage yml, node "data"
(seamlessClone_imp.cu:226-237).  Prints the line the reference prints with bSync=True
(seamlessClone_imp.cu:343-346).  Images may also be .bmp/.jpg/.png.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import capi, compare, ymlio


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="seamlessClone_main")
    ap.add_argument("src"); ap.add_argument("dst"); ap.add_argument("mask")
    ap.add_argument("centerX", type=int); ap.add_argument("centerY", type=int); ap.add_argument("gpu", type=int)
    ap.add_argument("--out", help="write the blended image (.bmp or .yml)")
    ap.add_argument("--method", default="auto", choices=["auto", "mg", "dst", "fft", "sor", "rbgs", "jacobi"],
                    help="auto (default): direct FFT solve (double) up to 720 unknowns per side (also elongated ROIs of at most 450 000 unknowns or at most 140 across), mg above; fft: the reference's default back-end (FFT-based direct solve, float32); mg: multigrid + float-table correction (the reference's arithmetic); dst: the reference's direct DST "
                         "solve on the fp64 matrix cores; sor / rbgs / jacobi: sweeps to a 2e-5 residual")
    ap.add_argument("--exact-tables", action="store_true", help="mg: return the exact solution of the 5-point system instead")
    ap.add_argument("--dump-rhs", metavar="DIR",
                    help="write the reference's SCDEBUG intermediates (seamlessClone_imp.cpp:2110-2117): DIR/ucMask0.yml (eroded ROI mask) and "
                         "DIR/g{0,1,2}.yml (right-hand side with the Dirichlet ring folded in, planes in the reference's R,G,B order) -- "
                         "what compare/vs.py:81-86 diffs against OpenCV's mod_diff{2,1,0}.yml")
    ap.add_argument("--reference-warmup", action="store_true",
                    help="clone twice in place like the reference binary (seamlessClone_imp.cu:303-318)")
    a = ap.parse_args(argv)
    src, dst, mask = compare._load(a.src), compare._load(a.dst), compare._load(a.mask)
    if mask.ndim == 3:
        mask = np.ascontiguousarray(mask[:, :, 0])
    print("mat shape: %d, %d, %d" % (src.shape[1], src.shape[0], 3))
    print("mat shape: %d, %d, %d" % (dst.shape[1], dst.shape[0], 3))
    print("mat shape: %d, %d, %d" % (mask.shape[1], mask.shape[0], 1))
    inst = capi.Instance(a.gpu)
    try:
        opts = {"method": {"auto": 5, "mg": 3, "dst": 4, "fft": 6, "sor": 2, "rbgs": 1, "jacobi": 0}[a.method],
                "reference_warmup": int(a.reference_warmup), "flags": capi.SC_FLAG_EXACT_TABLES if a.exact_tables else 0}
        if a.method not in ("auto", "mg", "dst", "fft"):
            opts.update(tol=2e-5, max_sweeps=1000000, check_every=64)
        inst.set_solver(**opts)
        body = np.array(dst, np.uint8, copy=True, order="C")
        inst.run(np.ascontiguousarray(src), body, np.ascontiguousarray(mask), a.centerX, a.centerY, sync=False)
        body2 = np.array(dst, np.uint8, copy=True, order="C")      # timed run after the warm-up, as the reference does
        sys.stdout.flush()       # the timed call (bSync = true) prints the reference's two lines from inside the library (C stdio)
        inst.run(np.ascontiguousarray(src), body2, np.ascontiguousarray(mask), a.centerX, a.centerY, sync=True)
        i = inst.info()
        print("device stages: %.3f msec (ROI %dx%d); transfers: H2D %.3f msec, D2H %.3f msec; solver %s: %d cycles/sweeps"
              % (i.ms_device_total, i.W, i.H, i.ms_h2d, i.ms_d2h, a.method, i.sweeps))
        if a.dump_rhs:
            dump_rhs(inst, a.dump_rhs, np.ascontiguousarray(src), np.ascontiguousarray(dst), np.ascontiguousarray(mask), a.centerX, a.centerY)
    finally:
        inst.destroy()
    if a.out:      # the timed call's image (the warm-up's holds the same bytes)
        if a.out.endswith((".yml", ".yml.gz")):
            ymlio.write_yml(a.out, body2, name="result")
        else:
            ymlio.write_bmp(a.out, body2)
    return 0


def folded_rhs(lap: np.ndarray, B: np.ndarray) -> np.ndarray:
    """The reference's `g` (seamlessClone_imp.cpp:1983-2009): the interior of the divergence `lap` with the destination's ring
    values subtracted on the four interior edges.  lap, B: [3][H][W] float32 planes (B, G, R) of the ROI; returns [3][H-2][W-2]."""
    g = lap[:, 1:-1, 1:-1].astype(np.float32, copy=True)
    g[:, :, 0] -= B[:, 1:-1, 0]
    g[:, 0, :] -= B[:, 0, 1:-1]
    g[:, :, -1] -= B[:, 1:-1, -1]
    g[:, -1, :] -= B[:, -1, 1:-1]
    return g


def dump_rhs(inst, out_dir, src, dst, mask, cx, cy) -> None:
    import os
    os.makedirs(out_dir, exist_ok=True)
    geo, M = inst.mask_stage(mask, cx, cy)
    ymlio.write_yml(os.path.join(out_dir, "ucMask0.yml"), M, name="ucMask0")
    geo, B, lap = inst.build_rhs(src, dst, mask, cx, cy)
    g = folded_rhs(lap, B)
    for ch in range(3):      # the reference's planes are R, G, B (seamlessClone_imp.cpp:374-376); ours follow the interleaved B, G, R
        ymlio.write_yml(os.path.join(out_dir, "g%d.yml" % ch), g[2 - ch], name="g%d" % ch)
    print("wrote %s/ucMask0.yml, g0.yml, g1.yml, g2.yml (%dx%d)" % (out_dir, g.shape[2], g.shape[1]))


if __name__ == "__main__":
    sys.exit(main())
