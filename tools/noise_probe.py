import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o
inst = capi.Instance(0)
rng = np.random.default_rng(99)
W, H = 300, 280; Hd, Wd = H + 64, W + 64
dst = rng.integers(0, 256, (Hd, Wd, 3), dtype=np.uint8); patch = rng.integers(0, 256, (H + 2, W + 2, 3), dtype=np.uint8); mask = np.full((H + 2, W + 2), 255, np.uint8)
want = o.seamless_clone(dst, patch, mask, Wd // 2, Hd // 2)
for ms in (2, 3, 4, 5):
    inst.set_solver(max_sweeps=ms, update_tol=1e-30)
    body = dst.copy(); inst.run(patch, body, mask, Wd // 2, Hd // 2, allow_not_converged=True); i = inst.info()
    print("noise cycles", i.sweeps, "last_update %.4f" % i.last_update, compare.format_stats(compare.image_diff_stats(want, body)))
