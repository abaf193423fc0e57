"""Fuzz of the size-class path (not part of the test suite): random batches whose members all differ in ROI size -- a base size
anywhere from 60 to 2300 a side, members within a few percent of it, so that most batches fall into one or two size classes --
through sc_hip_run_device_batch; every member against the float-table C port (+-1) and against its own solo run (byte-identical
whenever the group took the solo run's cycle count).  Rectangular / holed / elliptic masks, different positions.
python tests/tools/fuzz_classes.py [batches] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o, oracle_c as oc

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
oc.build()
inst = capi.Instance(0)
solo = capi.Instance(0)
solo.set_solver(method=capi.SC_METHOD_MULTIGRID)
nt = min(16, oc.max_threads())
members = shared = identical = compared = fails = 0
worst = 0.0
for bi in range(nb):
    base_w, base_h = int(rng.integers(60, 2300)), int(rng.integers(60, 2300))
    if bi % 4 == 0:
        base_h = base_w = int(rng.choice([100, 160, 195, 320, 520, 1010, 1040, 1090, 2040, 2110, 2120, 3100]))      # near class boundaries
    n = int(rng.integers(2, 7 if base_w * base_h > 2.5e6 else 12))
    spread = float(rng.choice([0.02, 0.05, 0.10, 0.3, 0.45]))
    sizes = [(max(12, int(base_w * (1 + rng.uniform(-spread, spread)))), max(12, int(base_h * (1 + rng.uniform(-spread, spread))))) for _ in range(n)]
    g, kinds = capi.plan_groups(sizes)          # keep the members of the largest size class: the call's statistics then describe THEIR launches
    from collections import Counter
    best = [q for q, c in Counter(gg for gg, kk in zip(g, kinds) if kk in (2, 3)).most_common(1)]
    if not best:
        continue
    sizes = [sz for sz, gg in zip(sizes, g) if gg == best[0]]
    n = len(sizes)
    g, kinds = capi.plan_groups(sizes)          # (3: on another hierarchy than the solo run's -- a small ROI, a leftover moved one level deeper)
    if len(set(g)) != 1:
        continue
    items = []
    for k, (W, H) in enumerate(sizes):
        dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=100 * bi + k, seed_patch=100 * bi + 50 + k, margin=24)
        mk = int(rng.integers(0, 3))
        if mk == 1:
            mask = mask.copy(); mask[H // 3:H // 3 + max(2, H // 9), W // 4:W // 4 + max(2, W // 5)] = 0
        elif mk == 2:
            yy, xx = np.mgrid[0:H + 2, 0:W + 2]
            mask = np.where(((yy - (H + 1) / 2) / (H / 2)) ** 2 + ((xx - (W + 1) / 2) / (W / 2)) ** 2 <= 1.0, 255, 0).astype(np.uint8)
        items.append((dst, patch, mask, cx + int(rng.integers(-6, 7)), cy + int(rng.integers(-6, 7))))
    jobs = capi.Pool.make_jobs(n); keep = []
    for j, (dst, patch, mask, cx, cy) in zip(jobs, items):
        f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
        keep.append((f, b0, b, m, dst.shape))
        j.face, j.face_cols, j.face_rows, j.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
        j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
        j.centerX, j.centerY, j.body_restore = cx, cy, b0
    inst.run_device_batch(jobs)
    info = inst.info()
    shared += info.group_members if info.group_ragged else 0
    gcycles = info.sweeps
    for k, ((f, b0, b, m, shape), it) in enumerate(zip(keep, items)):
        members += 1
        got = inst.from_device(b, shape)
        want = oc.seamless_clone(it[0], it[1], it[2], it[3], it[4], nthreads=nt, exact_den=False)
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        worst = max(worst, 100.0 * np.count_nonzero(d) / d.size)
        if d.max() > 1:
            fails += 1; print("batch", bi, "member", k, sizes[k], "max diff vs port", int(d.max()), flush=True)
        body = it[0].copy()
        solo.run(it[1], body, it[2], it[3], it[4])
        if not np.abs(body.astype(np.int16) - got.astype(np.int16)).max() <= 1:
            fails += 1; print("batch", bi, "member", k, sizes[k], "more than one grey level from its solo run", flush=True)
        if info.group_ragged and solo.info().sweeps == gcycles and solo.info().method == capi.SC_METHOD_MULTIGRID and kinds[k] == 2:
            compared += 1
            if np.array_equal(body, got):
                identical += 1
            elif capi.plan_size(*sizes[k])["conditional"]:      # the group's measured update decided the output's form, not this member's own
                print("batch", bi, "member", k, sizes[k], "(conditional) differs from its solo run in", int((body != got).sum()), "bytes", flush=True)
            else:
                fails += 1; print("batch", bi, "member", k, sizes[k], "differs from its solo run in", int((body != got).sum()), "bytes", flush=True)
    for kp in keep:
        for p in kp[:4]:
            inst.free(p)
    if bi % 10 == 9:
        print("...", bi + 1, "batches", members, "members", flush=True)
print("batches", nb, "members", members, "in the last size class of their call", shared, "compared byte for byte with their solo run", compared,
      "identical", identical, "fails", fails, "worst percent of channels off by one vs the port %.3f" % worst)
inst.destroy(); solo.destroy()
