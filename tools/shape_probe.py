import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
rng = np.random.default_rng(5)
for spec in sys.argv[1:]:
    W, H = (int(v) for v in spec.split("x"))
    Hd, Wd = H + 64, W + 64
    dst = rng.integers(60, 200, (Hd, Wd, 3), dtype=np.uint8); patch = rng.integers(60, 200, (H + 2, W + 2, 3), dtype=np.uint8)
    mask = np.full((H + 2, W + 2), 255, np.uint8)
    best = None
    for rep in range(3):
        body = dst.copy(); inst.run(patch, body, mask, Wd // 2, Hd // 2, allow_not_converged=True); i = inst.info()
        if best is None or i.ms_device_total < best[0]:
            best = (i.ms_device_total, i.ms_mask, i.ms_pre, i.ms_solve, i.ms_post, i.sweeps)
    print(spec, "device %.2f ms  mask %.2f pre %.2f solve %.2f post %.2f cycles %d  -> %.0f Mpix/s" % (*best, W * H / best[0] / 1e3), flush=True)
