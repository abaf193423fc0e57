import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np, bench
from seamlesscloneoptimization_amd import capi
pool = capi.Pool(0, 1, group=16, method=capi.SC_METHOD_MULTIGRID)
inst = pool.instances[0]
gen = bench.BatchSynth(2048, 1001)
n = 16
cj = pool.make_jobs(n)
for b in range(n):
    dst, patch, mask, cx, cy = gen.image(b)
    c = cj[b]
    c.face, c.face_cols, c.face_rows, c.face_step = inst.to_device(patch), patch.shape[1], patch.shape[0], 3 * patch.shape[1]
    c.body, c.body_cols, c.body_rows, c.body_step = inst.to_device(dst), dst.shape[1], dst.shape[0], 3 * dst.shape[1]
    c.mask, c.mask_cols, c.mask_rows, c.mask_step = inst.to_device(mask), mask.shape[1], mask.shape[0], mask.shape[1]
    c.centerX, c.centerY, c.body_restore = cx, cy, inst.to_device(dst)
for _ in range(3): pool.run(cj, device_resident=True)
for rep in range(3):
    print([round(inst.time_cycle0_form(k, 20) * 1e3, 1) for k in (3, 0, 1, 2)], flush=True)
