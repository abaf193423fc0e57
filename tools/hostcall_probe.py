"""GPU box: the host-image call (my_seamlessclone_api_imp_run on pageable numpy images) for one ROI size inside destinations of different widths
(whole-row copies when the ROI covers >= 3/4 of the row step, packed rows otherwise).  python tools/hostcall_probe.py [roi] [dst widths ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
widths = [int(a) for a in sys.argv[2:]] or [roi + 64, roi + 512, 2 * roi, 3 * roi]
rng = np.random.default_rng(1)
patch = rng.integers(0, 256, (roi + 2, roi + 2, 3), dtype=np.uint8)
mask = np.full((roi + 2, roi + 2), 255, np.uint8)
inst = capi.Instance(0)
inst.set_solver(flags=capi.SC_FLAG_NO_STAGE_MARKS)
for wd in widths:
    dst = np.clip(128.0 + rng.normal(0.0, 14.0, (roi + 64, wd, 3)), 0, 255).astype(np.uint8)
    body = dst.copy()
    cx, cy = wd // 2, (roi + 64) // 2
    t = []
    for k in range(14):
        body[...] = dst
        t0 = time.perf_counter(); inst.run(patch, body, mask, cx, cy); t.append((time.perf_counter() - t0) * 1e3)
    t = sorted(t[2:])
    i = inst.info()
    print(f"roi {roi}^2 in dst {wd} x {roi + 64}: call median {t[len(t)//2]:.3f} ms min {t[0]:.3f}  (stream {i.ms_call:.3f}: h2d {i.ms_h2d:.3f} device {i.ms_device_total:.3f} d2h {i.ms_d2h:.3f})", flush=True)
