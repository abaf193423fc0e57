"""GPU box: where a host-image call on a SMALL patch spends its time (the reference's own published sizes, 1600 x 898 destination): wall
time of my_seamlessclone_api_imp_run on pageable images with and without stage marks, its stream-side stages, and -- the floor of the
launch path -- the same clone on device-resident images (sc_hip_run_device, synchronous).  python tools/small_call_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi

rng = np.random.default_rng(3)
dst = np.clip(128.0 + rng.normal(0.0, 14.0, (898, 1600, 3)), 0, 255).astype(np.uint8)
for pw, ph in ((154, 100), (300, 194), (420, 300), (592, 592), (800, 800)):
    patch = rng.integers(0, 256, (ph, pw, 3), dtype=np.uint8)
    mask = np.full((ph, pw), 255, np.uint8)
    cx, cy = 800, 449
    row = {"patch": [pw, ph]}
    for name, flags in (("marks", 0), ("no_marks", capi.SC_FLAG_NO_STAGE_MARKS)):
        inst = capi.Instance(0); inst.set_solver(flags=flags)
        body = dst.copy()
        for _ in range(3):
            inst.run(patch, body, mask, cx, cy)
        ts = []
        for _ in range(40):
            body[...] = dst
            t0 = time.perf_counter(); inst.run(patch, body, mask, cx, cy); ts.append((time.perf_counter() - t0) * 1e3)
        i = inst.info(); ts.sort()
        row["host_call_" + name] = {"wall_ms_median": round(ts[20], 4), "min": round(ts[0], 4), "stream_ms": round(i.ms_call, 4), "h2d": round(i.ms_h2d, 4),
                                    "device": round(i.ms_device_total, 4), "d2h": round(i.ms_d2h, 4), "method": i.method}
        inst.destroy()
    inst = capi.Instance(0); inst.set_solver(flags=capi.SC_FLAG_NO_STAGE_MARKS)
    f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
    for _ in range(3):
        inst.run_device(f, patch.shape, b, dst.shape, m, mask.shape, cx, cy)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); inst.run_device(f, patch.shape, b, dst.shape, m, mask.shape, cx, cy); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    row["device_resident_call"] = {"wall_ms_median": round(ts[20], 4), "min": round(ts[0], 4), "device": round(inst.info().ms_device_total, 4)}
    inst.destroy()
    print(json.dumps(row), flush=True)
