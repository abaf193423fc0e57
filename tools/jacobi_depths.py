import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
for roi in (2048, 4096):
    rng = np.random.default_rng(1)
    U = rng.normal(100, 30, (3, roi, roi)).astype(np.float32); F = rng.normal(0, 10, (3, roi, roi)).astype(np.float32)
    inst.field_load(U, F)
    for spl, n in ((1, 100), (-1, 100), (2, 100), (4, 100), (8, 96)):
        ms = min(inst.field_time_sweeps(0, n, spl, 1.0) for _ in range(3))
        print(roi, "spl", spl, "%.1f us/sweep  %.0f GB/s" % (ms*1e3, 12.0*(roi-2)**2*3/ms/1e6), flush=True)
