"""GPU box: first call at new ROI sizes in the direct solve's range vs the repeated call (one instance, arena grown first)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
rng = np.random.default_rng(2)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 720)
big = hi + 2
noise_p = rng.integers(0, 256, (big, big, 3), dtype=np.uint8)
noise_d = np.clip(128.0 + rng.normal(0.0, 14.0, (big + 64, big + 64, 3)), 0, 255).astype(np.uint8)
mask = np.full((big, big), 255, np.uint8)
inst = capi.Instance(0)
dev = (inst.to_device(noise_p), inst.to_device(noise_d), inst.to_device(noise_d), inst.to_device(mask))
def call(pw, ph):
    inst.copy_d2d_async(dev[1], dev[2], noise_d.nbytes); inst.sync()
    t0 = time.perf_counter()
    rc = inst.L.sc_hip_run_device(inst.h, dev[0], pw, ph, 3 * big, dev[1], pw + 64, ph + 64, 3 * (big + 64), dev[3], pw, ph, big, (pw + 64) // 2, (ph + 64) // 2, True)
    return (time.perf_counter() - t0) * 1e3
call(big, big); call(big, big)
rows = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    pw, ph = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    f = call(pw, ph); s = min(call(pw, ph), call(pw, ph))
    rows.append((f, s)); print(f"{pw}x{ph} first {f:.3f} steady {s:.3f} ratio {f/s:.2f}", flush=True)
r = sorted(f / s for f, s in rows); print("median ratio", r[len(r)//2], "median first", sorted(f for f, _ in rows)[len(rows)//2], "steady", sorted(s for _, s in rows)[len(rows)//2])
