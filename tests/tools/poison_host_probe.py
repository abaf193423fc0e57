"""Diagnostic: host-image calls with SC_FLAG_POISON_ARENA (device blocks handed out unzeroed AND fresh pinned staging filled with 0xFF)
against the same calls without it, and against the port."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o, oracle_c as oc
oc.build()
sizes = [(300, 310), (1003, 1010), (318, 333), (640, 480), (90, 70), (340, 305), (154, 100), (592, 592), (420, 300)]
bad = 0
for method in (capi.SC_METHOD_MULTIGRID, capi.SC_METHOD_AUTO):
    for k, (W, H) in enumerate(sizes):
        dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=40 + k, seed_patch=90 + k, margin=36)
        want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=16, exact_den=False)
        res = []
        for flags in (0, capi.SC_FLAG_POISON_ARENA):
            inst = capi.Instance(0); inst.set_solver(method=method, flags=flags)
            outs = []
            for rep in range(2):
                b = dst.copy(); inst.run(patch, b, mask, cx, cy); outs.append(b)
            inst.destroy(); res.append(outs)
        for rep in range(2):
            for name, a in (("plain", res[0][rep]), ("poisoned", res[1][rep])):
                d = np.abs(a.astype(np.int16) - want.astype(np.int16)).max(axis=2)
                if d.max() > 1:
                    ys, xs = np.nonzero(d > 1); bad += 1
                    print("method", method, (W, H), name, "rep", rep, "off the port in", len(ys), "pixels max", int(d.max()), "box x", int(xs.min()), int(xs.max()), "y", int(ys.min()), int(ys.max()), "roi at x", cx - (W + 2) // 2, "y", cy - (H + 2) // 2, flush=True)
            if not np.array_equal(res[0][rep], res[1][rep]):
                print("method", method, (W, H), "rep", rep, "poisoned differs from plain in", int((res[0][rep] != res[1][rep]).any(axis=2).sum()), "pixels", flush=True); bad += 1
print("bad", bad)
