"""Long-running fuzz of the multigrid range (not part of the test suite): random ROI shapes from the AUTO crossover up to ~2600 a
side incl. elongated ones, random rectangular / elliptic / speckled masks, every mask cloned twice (the second call predicts the
remembered box: erode inside the pre-process tiles), every result against the C port with float tables.
python tests/tools/fuzz_mg.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
inst = capi.Instance(0)
nt = min(16, oc.max_threads())
worst = (0.0, None); fails = 0
for case in range(n):
    kind = case % 4
    if kind == 0: W, H = int(rng.integers(725, 2600)), int(rng.integers(725, 2600))
    elif kind == 1: W, H = int(rng.integers(1000, 4000)), int(rng.integers(40, 400))
    elif kind == 2: W, H = int(rng.integers(40, 400)), int(rng.integers(1000, 4000))
    else: W, H = int(rng.choice([1015, 1016, 1017, 1018, 1019, 1020, 2031, 2040, 2045, 2046, 2047, 2048, 2049, 2050, 2064, 2065, 2070])), int(rng.integers(725, 2100))
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=case, seed_patch=1000 + case, margin=16)
    mk = int(rng.integers(0, 4))
    if mk == 1:
        mask = np.zeros_like(mask); mask[int(rng.integers(1, 9)):H - int(rng.integers(0, 9)), int(rng.integers(1, 9)):W - int(rng.integers(0, 9))] = 255
    elif mk == 2:
        yy, xx = np.mgrid[0:H + 2, 0:W + 2]
        mask = np.where(((yy - H / 2) / (H / 2 - 1)) ** 2 + ((xx - W / 2) / (W / 2 - 1)) ** 2 <= 1.0, 255, 0).astype(np.uint8)
    elif mk == 3:
        mask = mask.copy(); mask[rng.integers(0, H + 2, 30), rng.integers(0, W + 2, 30)] = rng.integers(0, 255, 30)
    try:
        want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=nt, exact_den=False)
    except Exception as e:
        print("case", case, W, H, mk, "oracle rejected:", e, flush=True); continue
    outs = []
    for rep in range(2):
        body = dst.copy()
        try:
            inst.run(patch, body, mask, cx, cy)
        except capi.SeamlessCloneError as e:
            print("case", case, W, H, mk, "GPU error", e, flush=True); fails += 1; break
        outs.append(body)
    if len(outs) < 2: continue
    if not np.array_equal(outs[0], outs[1]): print("case", case, W, H, mk, "second call differs from the first", flush=True); fails += 1
    s = compare.image_diff_stats(want, outs[1])
    if s["max"] > 1: print("case", case, W, H, mk, "FAIL", compare.format_stats(s), flush=True); fails += 1
    if s["percent"] > worst[0]: worst = (s["percent"], (case, W, H, mk))
    if case % 10 == 9: print("...", case + 1, "cases, fails", fails, "worst percent", worst, "cycles", inst.info().sweeps, flush=True)
print("cases", n, "fails", fails, "worst percent", worst, flush=True)
