"""What the storage formats of the multigrid fast path cost in off-by-one channels (against the float-table C port) and buy in
device time: default (16-bit fixed-point field between the level-0 launches, float16 level 1), SC_FLAG_FLOAT_FIELD (float
field, float16 level 1), SC_FLAG_FLOAT_L1 (both float).  Run on the GPU box: python tests/tools/field_format_probe.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc
inst = capi.Instance(0)
inst.set_solver(method=capi.SC_METHOD_MULTIGRID)
for (W, H, sd, sp, kind) in [(2048, 2048, 1001, 2002, "std"), (2048, 2048, 7, 8, "std"), (1500, 1100, 3, 4, "std"), (1024, 1024, 5, 6, "std"),
                             (2048, 2048, 1, 2, "noise"), (1777, 1333, 9, 10, "bw"), (600, 400, 11, 12, "std"), (4096, 4096, 13, 14, "std"), (3000, 500, 15, 16, "std")]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=sd, seed_patch=sp, margin=64)
    rng = np.random.default_rng(sd)
    if kind == "noise":
        patch = rng.integers(0, 256, patch.shape, dtype=np.uint8); dst = rng.integers(0, 256, dst.shape, dtype=np.uint8)
    if kind == "bw":
        patch = (rng.integers(0, 2, patch.shape) * 255).astype(np.uint8)
    want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=16, exact_den=False)
    row = [f"{W}x{H} {kind}"]
    for name, fl in (("q16", 0), ("float field", capi.SC_FLAG_FLOAT_FIELD), ("float L1", capi.SC_FLAG_FLOAT_L1)):
        inst.set_solver(flags=fl)
        best = 1e9
        for _ in range(4):
            body = dst.copy(); inst.run(patch, body, mask, cx, cy, sync=True)
            best = min(best, inst.info().ms_device_total)
        i = inst.info(); s = compare.image_diff_stats(want, body)
        row.append(f"{name}: cycles {i.sweeps} dev {best:.3f} ms max {s['max']} pct {s['percent']:.3f}")
    print(" | ".join(row), flush=True)
