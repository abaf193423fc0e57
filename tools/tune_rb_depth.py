import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
for roi in (2048, 4096):
    rng = np.random.default_rng(1)
    U = rng.normal(100, 30, (3, roi, roi)).astype(np.float32); F = rng.normal(0, 10, (3, roi, roi)).astype(np.float32)
    inst.field_load(U, F)
    for spl in (2, 3, 4):
        ms = min(inst.field_time_sweeps(capi.SC_METHOD_SOR, 60, spl, 1.7) for _ in range(2))
        print(roi, "SOR depth", spl, "%.1f us/launch  %.2f us/sweep  %.0f GB/s eff" % (ms*1e3, ms*1e3/spl, 12.0*(roi-2)**2*3*spl/ms/1e6), flush=True)
