"""GPU box: one 2048^2 clone on device-resident images at four ROI columns (3 x column mod 4 = the output rows' alignment):
device time box to box and the time from the solve's end to the clone's end (the splice)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o
roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=64)
inst = capi.Instance(0)
f, b0, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(mask)
out = {}
for shift in range(4):
    b = inst.to_device(dst)
    ts = []
    for rep in range(12):
        inst.run_device(f, patch.shape, b, dst.shape, m, mask.shape, cx + shift, cy)
        i = inst.info(); ts.append((i.ms_device_total, i.ms_solve, i.ms_post))
    ts.sort()
    out["shift_%d" % shift] = {"ltx": int(cx + shift - (roi + 2) // 2), "device_ms": round(ts[6][0], 4), "solve": round(ts[6][1], 4), "post": round(ts[6][2], 4)}
    inst.free(b)
print(json.dumps(out))
