"""GPU box: throughput of a batch whose members all have DIFFERENT ROI sizes (what real clones produce) against the same number of
same-size clones, through the native pool (device-resident images, destinations refreshed inside the step).

    python tools/mixed_probe.py [--lo 1000 --hi 1100] [--n 64] [--streams 2] [--group 16] [--reps 8] [--seed 2025]

Lines: mixed sizes through the pool as configured (size classes share launches), the same list with groups of ONE on 8 streams
(round 4's behaviour for such a batch), n same-size clones of the list's mean size (the ceiling)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seamlesscloneoptimization_amd import capi  # noqa: E402


def make_images(hi, seed):
    rng = np.random.default_rng(seed)
    Hd = Wd = hi + 64
    yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
    dst = np.clip((128.0 + 60.0 * np.sin(2 * np.pi * xx / Wd) * np.cos(2 * np.pi * yy / Hd))[:, :, None] +
                  12.0 * rng.standard_normal((Hd, Wd, 3), dtype=np.float32), 0, 255).astype(np.uint8)
    xx = np.arange(hi + 2, dtype=np.float32)[None, :, None]
    patch = np.clip(110.0 + 50.0 * np.cos(3 * np.pi * xx / hi) + 20.0 * rng.standard_normal((hi + 2, hi + 2, 3), dtype=np.float32), 0, 255).astype(np.uint8)
    return dst, patch


def jobs_for(pool, sizes, dst, patch):
    inst = pool.instances[0]
    jobs = pool.make_jobs(len(sizes)); keep = []
    d_dst0 = inst.to_device(dst)
    for j, (W, H) in zip(jobs, sizes):
        p = np.ascontiguousarray(patch[:H + 2, :W + 2])
        m = np.full((H + 2, W + 2), 255, np.uint8)
        f, b, dm = inst.to_device(p), inst.to_device(dst), inst.to_device(m)
        keep += [f, b, dm]
        j.face, j.face_cols, j.face_rows, j.face_step = f, W + 2, H + 2, 3 * (W + 2)
        j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        j.mask, j.mask_cols, j.mask_rows, j.mask_step = dm, W + 2, H + 2, W + 2
        j.centerX, j.centerY, j.body_restore = dst.shape[1] // 2, dst.shape[0] // 2, d_dst0
    keep.append(d_dst0)
    return jobs, keep


def time_pool(streams, group, sizes, dst, patch, reps, unseen=False):
    pool = capi.Pool(0, streams=streams, group=group)
    try:
        jobs, keep = jobs_for(pool, sizes, dst, patch)
        pool.run(jobs, device_resident=True)
        pool.run(jobs, device_resident=True)
        ts = []
        for _ in range(reps):
            if unseen:
                capi.plan_cache_clear()      # as if no size had ever been planned: every member pays its plan and its host tables again
            t0 = time.perf_counter()
            pool.run(jobs, device_resident=True)
            ts.append(time.perf_counter() - t0)
        cycles = max(i.info().sweeps for i in pool.instances)
        shared = max(i.info().group_members for i in pool.instances)
        for p in keep:
            pool.instances[0].free(p)
    finally:
        pool.close()
    ts.sort()
    mpix = sum(w * h for w, h in sizes) / 1e6
    return {"ms_median": round(ts[len(ts) // 2] * 1e3, 3), "ms_min": round(ts[0] * 1e3, 3), "Gpix_per_s": round(mpix / ts[len(ts) // 2] / 1e3, 2),
            "cycles": cycles, "largest_shared_group": shared}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=1000); ap.add_argument("--hi", type=int, default=1100)
    ap.add_argument("--n", type=int, default=64); ap.add_argument("--streams", type=int, default=2); ap.add_argument("--group", type=int, default=16)
    ap.add_argument("--reps", type=int, default=8); ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--legs", default="mixed,ones,same", help="which of the three measurements to run")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    sizes = [(int(rng.integers(a.lo, a.hi + 1)), int(rng.integers(a.lo, a.hi + 1))) for _ in range(a.n)]
    g, k = capi.plan_groups_pool(sizes, a.group, a.streams)      # (--group 0: SC_POOL_GROUP_AUTO)
    from collections import Counter
    dst, patch = make_images(a.hi, a.seed)
    mean = int(round(np.sqrt(np.mean([w * h for w, h in sizes]))))
    out = {"range": [a.lo, a.hi], "n": a.n, "streams": a.streams, "group": a.group,
           "planned_groups": sorted(Counter(g).values(), reverse=True), "kinds": dict(Counter(k))}
    legs = a.legs.split(",")
    if "mixed" in legs: out["mixed_sizes"] = time_pool(a.streams, a.group, sizes, dst, patch, a.reps)
    if "mixed" in legs: out["mixed_sizes_planner_memo_cleared_every_step"] = time_pool(a.streams, a.group, sizes, dst, patch, a.reps, unseen=True)
    if "ones" in legs: out["mixed_sizes_groups_of_one_8_streams"] = time_pool(8, 1, sizes, dst, patch, a.reps)
    if "same" in legs: out["same_size_%d" % mean] = time_pool(a.streams, a.group, [(mean, mean)] * a.n, dst, patch, a.reps)
    if "mixed" in legs and "same" in legs:
        out["ratio_to_same_size"] = round(out["mixed_sizes"]["Gpix_per_s"] / out["same_size_%d" % mean]["Gpix_per_s"], 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
