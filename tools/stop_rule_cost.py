"""GPU box: what a stricter stop rule would buy and cost (round-4 review, item 5's option).  update_tol 0.25 (default) against 0.125:
cycles, device time and the share of channels off by one against the float-table port -- full-range noise (300x280, 1024x700), and a
smooth 2048^2 clone (the bench workload's kind of image).  python tools/stop_rule_cost.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o, oracle_c as oc

oc.build()
rng = np.random.default_rng(11)
cases = []
for W, H in ((300, 280), (1024, 700)):
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=24)
    cases.append(("noise_%dx%d" % (W, H), rng.integers(0, 256, dst.shape, dtype=np.uint8), rng.integers(0, 256, patch.shape, dtype=np.uint8), mask, cx, cy))
dst, patch, mask, cx, cy = o.synth_inputs(2048, 2048, margin=24)
cases.append(("smooth_2048x2048", dst, patch, mask, cx, cy))
out = []
for name, dst, patch, mask, cx, cy in cases:
    want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=min(16, oc.max_threads()), exact_den=False)
    row = {"case": name}
    for tol in (0.25, 0.125, 0.0625):
        inst = capi.Instance(0)
        inst.set_solver(method=capi.SC_METHOD_MULTIGRID, update_tol=tol)
        body = dst.copy(); inst.run(patch, body, mask, cx, cy)
        ms = []
        for _ in range(5):
            body = dst.copy(); inst.run(patch, body, mask, cx, cy); ms.append(inst.info().ms_device_total)
        d = np.abs(body.astype(np.int16) - want.astype(np.int16))
        row["update_tol_%g" % tol] = {"cycles": inst.info().sweeps, "device_ms": round(sorted(ms)[2], 4), "max": int(d.max()),
                                      "percent_off_by_one": round(100.0 * np.count_nonzero(d) / d.size, 4)}
        inst.destroy()
    out.append(row)
    print(json.dumps(row), flush=True)
