"""Bottom solve on the matrix cores (default) against SC_FLAG_BOTTOM_F32: cycles, last correction, off-by-one share against the
numpy oracle for a few stress inputs.  python tests/tools/bottom_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o

inst = capi.Instance(0)
rng = np.random.default_rng(99)
cases = []
W, H = 300, 280
Hd, Wd = H + 64, W + 64
cases.append(("noise 300x280", rng.integers(0, 256, (Hd, Wd, 3), dtype=np.uint8), rng.integers(0, 256, (H + 2, W + 2, 3), dtype=np.uint8)))
for (W, H) in ((700, 560), (1200, 900)):
    d, p, m, cx, cy = o.synth_inputs(W, H, margin=64)
    cases.append((f"synth {W}x{H}", d, p))
    cases.append((f"noise {W}x{H}", rng.integers(0, 256, d.shape, dtype=np.uint8), rng.integers(0, 256, p.shape, dtype=np.uint8)))
for name, dst, patch in cases:
    mask = np.full(patch.shape[:2], 255, np.uint8)
    cx, cy = dst.shape[1] // 2, dst.shape[0] // 2
    want = o.seamless_clone(dst, patch, mask, cx, cy, float_tables=True)
    for flags in (0, capi.SC_FLAG_LEGACY_PATHS):      # with legacy_paths = SC_LEGACY_BOTTOM_F32 (set below)
        inst.set_solver(method=capi.SC_METHOD_MULTIGRID, flags=flags, legacy_paths=capi.SC_LEGACY_BOTTOM_F32)
        body = dst.copy()
        inst.run(patch, body, mask, cx, cy, allow_not_converged=True)
        i = inst.info()
        d = np.abs(body.astype(int) - want.astype(int))
        print(f"{name:18s} flags {flags:5d}: cycles {i.sweeps} last_update {i.last_update:.4f} max {d.max()} differing {100.0 * (d > 0).mean():.4f} %", flush=True)
