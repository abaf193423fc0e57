"""One GROUP of same-size clones at a time on one stream (sc_hip_run_device_batch) for a kernel-trace timeline:
python tools/group_trace.py [roi] [group] [n]   (fold with tools/trace_timeline.py: a group starts at k_mask_bbox_group)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o
roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
group = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pool = capi.Pool(0, 1, group=group)
inst = pool.instances[0]
jobs = pool.make_jobs(group)
for k, j in enumerate(jobs):
    dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, seed_dst=11 + k, seed_patch=31 + k, margin=256)
    f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
    j.face, j.face_cols, j.face_rows, j.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
    j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
    j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
    j.centerX, j.centerY, j.body_restore = cx, cy, b0
import time
for i in range(n):
    t0 = time.perf_counter(); pool.run(jobs, device_resident=True); dt = time.perf_counter() - t0
print("group of %d at %d^2: %.3f ms per group, %.1f Mpix/s, cycles %d" % (group, roi, dt * 1e3, group * roi * roi / dt / 1e6, inst.info().sweeps))
