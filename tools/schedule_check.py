import sys, os; sys.path.insert(0, "/root/repo")
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc
inst = capi.Instance(0)
rng = np.random.default_rng(99)
W, H = 300, 280; Hd, Wd = H + 64, W + 64
dst = rng.integers(0, 256, (Hd, Wd, 3), dtype=np.uint8); patch = rng.integers(0, 256, (H + 2, W + 2, 3), dtype=np.uint8); mask = np.full((H + 2, W + 2), 255, np.uint8)
want = o.seamless_clone(dst, patch, mask, Wd // 2, Hd // 2)
body = dst.copy(); inst.run(patch, body, mask, Wd // 2, Hd // 2); i = inst.info()
print("noise 300x280: cycles", i.sweeps, "last_update %.4f" % i.last_update, compare.format_stats(compare.image_diff_stats(want, body)))
for (W, H) in [(298, 192), (1024, 1024), (2048, 2048), (4096, 4096)]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=64)
    want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=16, exact_den=True)
    body = dst.copy(); inst.run(patch, body, mask, cx, cy); i = inst.info()
    print(W, H, "cycles", i.sweeps, "last_update %.4f" % i.last_update, "device %.3f ms" % i.ms_device_total, compare.format_stats(compare.image_diff_stats(want, body)), flush=True)
