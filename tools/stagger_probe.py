"""Do two streams of groups run faster when they are out of phase?  In `bench.py` both workers start a step together and run the
same launches at the same moments (kernel trace: the two streams' level-0 launches start within 10 us of each other), so a
stream's latency-bound coarse levels only ever overlap the other stream's coarse levels.  Here: two one-stream pools, one Python
thread each (ctypes releases the GIL), each running its group of 16 x roi^2 clones `steps` times without a barrier in between,
the second thread starting `offset` ms after the first.  Prints Mpix/s per offset.
usage: python tools/stagger_probe.py [roi] [steps] [offset_ms ...]"""
import sys, os, time, threading, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
import bench

roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
offsets = [float(a) for a in sys.argv[3:]] or [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0]
G = 16
pools = [capi.Pool(0, 1, group=G, method=capi.SC_METHOD_MULTIGRID) for _ in range(2)]
jobsets = []
for k, pool in enumerate(pools):
    inst = pool.instances[0]
    cj = pool.make_jobs(G)
    keep = []
    for b in range(G):
        dst, patch, mask, cx, cy = bench.synth(roi, k * G + b)
        f, b0, bb, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
        keep.append((f, b0, bb, m))
        c = cj[b]
        c.face, c.face_cols, c.face_rows, c.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
        c.body, c.body_cols, c.body_rows, c.body_step = bb, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        c.mask, c.mask_cols, c.mask_rows, c.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
        c.centerX, c.centerY, c.body_restore = cx, cy, b0
    jobsets.append((cj, keep))
    pool.run(cj, device_resident=True)
    pool.run(cj, device_resident=True)

def loop(k, n, delay):
    if delay > 0:
        time.sleep(delay)
    for _ in range(n):
        pools[k].run(jobsets[k][0], device_resident=True)

for off in offsets:
    best = 0.0
    for rep in range(2):
        ts = [threading.Thread(target=loop, args=(k, steps, off * 1e-3 if k == 1 else 0.0)) for k in range(2)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        best = max(best, 2 * G * steps * roi * roi / dt / 1e6)
    print(json.dumps({"offset_ms": off, "Mpix/s": round(best), "ms_per_step": round(2 * G * roi * roi / best / 1e3, 3)}), flush=True)
