"""Shape fuzz of the grouped path (not part of the test suite): random groups of same-size clones (tiny, thin, level-boundary
and ordinary ROI sizes; rectangular / elliptic / speckled masks; different positions), every member against the numpy oracle.
python tools/fuzz_groups.py [groups] [seed] [big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o

ngroups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
big = len(sys.argv) > 3 and sys.argv[3] == "big"       # every third group with an ROI of 900..2300 x 700..2100
pool = capi.Pool(0, 1, group=64)
inst = pool.instances[0]
fails = 0; members = 0; worst = 0.0
for gi in range(ngroups):
    kind = gi % 5
    if kind == 0: W, H = int(rng.integers(3, 24)), int(rng.integers(3, 24))
    elif kind == 1: W, H = int(rng.integers(200, 900)), int(rng.integers(3, 24))
    elif kind == 2: W, H = int(rng.integers(3, 24)), int(rng.integers(200, 600))
    elif kind == 3: W, H = int(rng.choice([63, 64, 65, 126, 127, 128, 129, 254, 255, 256, 257])), int(rng.choice([63, 64, 65, 66, 127, 128, 129, 130]))
    else: W, H = int(rng.integers(24, 600)), int(rng.integers(24, 400))
    if big and gi % 3 == 0: W, H = int(rng.integers(900, 2300)), int(rng.integers(700, 2100))   # interior-wave paths, many rounds
    n = int(rng.integers(2, 6 if (big and gi % 3 == 0) else 10))
    items = []
    for k in range(n):
        dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=1000 * gi + k, seed_patch=1000 * gi + 500 + k, margin=24)
        mk = int(rng.integers(0, 3))
        if mk == 1 and W > 12 and H > 12:            # holes and speckles that keep the bounding box
            mask = mask.copy(); mask[rng.integers(3, H - 2, 5), rng.integers(3, W - 2, 5)] = rng.integers(0, 255, 5)
        elif mk == 2 and W > 12 and H > 12:
            yy, xx = np.mgrid[0:H + 2, 0:W + 2]
            mask = np.where(((yy - (H + 1) / 2) / (H / 2)) ** 2 + ((xx - (W + 1) / 2) / (W / 2)) ** 2 <= 1.0, 255, 0).astype(np.uint8)
        items.append((dst, patch, mask, cx + int(rng.integers(-8, 9)), cy + int(rng.integers(-8, 9))))
    jobs = pool.make_jobs(n); keep = []
    for j, (dst, patch, mask, cx, cy) in zip(jobs, items):
        f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
        keep.append((f, b0, b, m, dst.shape))
        j.face, j.face_cols, j.face_rows, j.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
        j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
        j.centerX, j.centerY, j.body_restore = cx, cy, b0
    try:
        pool.run(jobs, device_resident=True)
        err = None
    except capi.SeamlessCloneError as e:
        err = e
    for k, ((f, b0, b, m, shape), it) in enumerate(zip(keep, items)):
        members += 1
        try:
            want = o.seamless_clone(*it, float_tables=True)
        except Exception:
            want = None
        if want is None or jobs[k].rc not in (0, capi.SC_ERR_NOT_CONVERGED):
            if (want is None) != (jobs[k].rc not in (0, capi.SC_ERR_NOT_CONVERGED)):
                fails += 1; print("group", gi, "member", k, "W,H", W, H, "rc", jobs[k].rc, "oracle", "rejects" if want is None else "accepts")
            continue
        got = inst.from_device(b, shape)
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        if d.max() > 1:
            fails += 1; print("group", gi, "member", k, "W,H", W, H, "n", n, "max diff", d.max())
        worst = max(worst, 100.0 * np.count_nonzero(d) / d.size)
    for kp in keep:
        for p in kp[:4]: inst.free(p)
print("groups", ngroups, "members", members, "fails", fails, "worst percent of differing channels %.3f" % worst)
