"""A/B of sc_solver_opts variants on the bench's timed step, in ONE process on one box (boxes differ by +-2 %):
python tools/ab_step.py "<name>=<field>:<value>[,<field>:<value>...]" ...      e.g.  base=flags:0 f32=flags:2050
Fields: any sc_solver_opts member; `reserved0` sets reserved[0]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
import bench

roi, batch, streams, group, steps = 2048, 32, 2, 16, 12
variants = []
for a in sys.argv[1:]:
    name, spec = a.split("=", 1)
    variants.append((name, [(f.split(":")[0], int(f.split(":")[1])) for f in spec.split(",") if f]))
pool = capi.Pool(0, streams, group=group, method=capi.SC_METHOD_MULTIGRID)
inst = pool.instances[0]
gen = bench.BatchSynth(roi, 1001)
cjobs = pool.make_jobs(batch)
for b in range(batch):
    dst, patch, mask, cx, cy = gen.image(b)
    c = cjobs[b]
    c.face, c.face_cols, c.face_rows, c.face_step = inst.to_device(patch), patch.shape[1], patch.shape[0], 3 * patch.shape[1]
    c.body, c.body_cols, c.body_rows, c.body_step = inst.to_device(dst), dst.shape[1], dst.shape[0], 3 * dst.shape[1]
    c.mask, c.mask_cols, c.mask_rows, c.mask_step = inst.to_device(mask), mask.shape[1], mask.shape[0], mask.shape[1]
    c.centerX, c.centerY, c.body_restore = cx, cy, inst.to_device(dst)


def apply(fields):
    o = inst.get_solver()
    o.reserved[0] = 0
    o.flags = 0
    for k, v in fields:
        if k == "reserved0":
            o.reserved[0] = v
        else:
            setattr(o, k, v)
    import ctypes as C
    assert pool.L.sc_hip_pool_set_solver(pool.h, C.byref(o)) == 0


for rep in range(3):
    for name, fields in variants:
        apply(fields)
        for _ in range(3):
            pool.run(cjobs, device_resident=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            pool.run(cjobs, device_resident=True)
        dt = (time.perf_counter() - t0) / steps
        print(f"rep {rep} {name:12s} {dt * 1e3:.4f} ms/step  {roi * roi * batch / dt / 1e6:.0f} Mpix/s  cycles {max(i.info().sweeps for i in pool.instances)}", flush=True)
