"""GPU box: the bench's new_size protocol with every call printed that is among the five slowest first calls of its range:
size, first / steady ms, whether the arena grew (sc_run_info.device_bytes), which method ran."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
big = 2402; dstw = big + 64
noise_p = rng.integers(0, 256, (big, big, 3), dtype=np.uint8)
noise_d = np.clip(128.0 + rng.normal(0.0, 14.0, (dstw, dstw, 3)), 0, 255).astype(np.uint8)
mask = np.full((big, big), 255, np.uint8)
inst = capi.Instance(0)
dev = (inst.to_device(noise_p), inst.to_device(noise_d), inst.to_device(noise_d), inst.to_device(mask))
def call(pw, ph):
    inst.copy_d2d_async(dev[1], dev[2], noise_d.nbytes); inst.sync()
    t0 = time.perf_counter()
    inst.L.sc_hip_run_device(inst.h, dev[0], pw, ph, 3 * big, dev[1], pw + 64, ph + 64, 3 * dstw, dev[3], pw, ph, big, (pw + 64) // 2, (ph + 64) // 2, True)
    return (time.perf_counter() - t0) * 1e3
call(2402, 2402); call(902, 902); call(capi.SC_AUTO_DIRECT_MAX + 2, capi.SC_AUTO_DIRECT_MAX + 2)
for name, lo, hi in (("roi_100_720", 100, capi.SC_AUTO_DIRECT_MAX), ("roi_1000_2400", 1000, 2400)):
    rows = []
    for k in range(64):
        pw, ph = int(rng.integers(lo, hi + 1)) + 2, int(rng.integers(lo, hi + 1)) + 2
        b0 = inst.info().device_bytes
        f = call(pw, ph); i = inst.info(); grew = i.device_bytes - b0
        s = min(call(pw, ph), call(pw, ph))
        rows.append((f / s, pw, ph, f, s, grew, i.method, k))
    rows.sort(reverse=True)
    r = sorted(x[0] for x in rows)
    print(name, "median %.3f p95 %.3f max %.3f" % (r[32], r[60], r[-1]))
    for x in rows[:6]:
        print("   ratio %.2f  %dx%d first %.3f steady %.3f arena grew by %d bytes method %d call #%d" % x)
