"""GPU box: host cost and total time of ONE sc_hip_run_device_batch call on one instance -- a size class against a same-size group.
    python tools/group_host_probe.py [--lo 300 --hi 340] [--n 16] [--reps 30]
enqueue_ms = until the (asynchronous) call returns, total_ms = until the instance's stream has drained."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from seamlesscloneoptimization_amd import capi  # noqa: E402
from mixed_probe import make_images, jobs_for  # noqa: E402


class OnePool:      # jobs_for wants .instances / .make_jobs
    def __init__(self, inst):
        self.instances = [inst]
    make_jobs = staticmethod(capi.Pool.make_jobs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=300); ap.add_argument("--hi", type=int, default=340)
    ap.add_argument("--n", type=int, default=16); ap.add_argument("--reps", type=int, default=30); ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--only", default="", help="class | same: run that variant alone (timelines)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dst, patch = make_images(a.hi, a.seed)
    out = {"range": [a.lo, a.hi], "n": a.n}
    cands = [(int(rng.integers(a.lo, a.hi + 1)), int(rng.integers(a.lo, a.hi + 1))) for _ in range(400)]
    g, k = capi.plan_groups(cands)
    from collections import Counter
    best = Counter(g).most_common(1)[0][0]
    sizes = [s for s, gg in zip(cands, g) if gg == best][:a.n]
    mean = int(round(np.sqrt(np.mean([w * h for w, h in sizes]))))
    for name, sz in (("size_class", sizes), ("same_size_%d" % mean, [(mean, mean)] * len(sizes))):
        if a.only and not name.startswith("size_class" if a.only == "class" else "same"):
            continue
        inst = capi.Instance(0)
        jobs, keep = jobs_for(OnePool(inst), sz, dst, patch)
        for _ in range(3):
            inst.run_device_batch(jobs)
        enq, tot = [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            inst.run_device_batch(jobs, sync=False)
            t1 = time.perf_counter()
            inst.sync()
            t2 = time.perf_counter()
            enq.append(t1 - t0); tot.append(t2 - t0)
        i = inst.info()
        enq.sort(); tot.sort()
        out[name] = {"members": len(sz), "shared": i.group_members, "ragged": i.group_ragged, "enqueue_ms": round(enq[len(enq) // 2] * 1e3, 3),
                     "total_ms": round(tot[len(tot) // 2] * 1e3, 3), "Gpix_per_s": round(sum(w * h for w, h in sz) / tot[len(tot) // 2] / 1e9, 2)}
        for p in keep:
            inst.free(p)
        inst.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
