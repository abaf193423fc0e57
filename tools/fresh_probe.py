"""GPU box: the first call of a fresh instance at one size, wall time (new_size leg's protocol)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4098
rng = np.random.default_rng(5)
p4 = rng.integers(0, 256, (n, n, 3), dtype=np.uint8)
d4 = np.clip(128.0 + rng.normal(0.0, 14.0, (n + 64, n + 64, 3)), 0, 255).astype(np.uint8)
m4 = np.full((n, n), 255, np.uint8)
inst = capi.Instance(0)
dev = (inst.to_device(p4), inst.to_device(d4), inst.to_device(d4), inst.to_device(m4))
for k in range(3):
    inst.copy_d2d_async(dev[1], dev[2], d4.nbytes); inst.sync()
    t0 = time.perf_counter()
    rc = inst.L.sc_hip_run_device(inst.h, dev[0], n, n, 3 * n, dev[1], n + 64, n + 64, 3 * (n + 64), dev[3], n, n, n, (n + 64) // 2, (n + 64) // 2, True)
    print("call", k, round((time.perf_counter() - t0) * 1e3, 3), "ms rc", rc, flush=True)
