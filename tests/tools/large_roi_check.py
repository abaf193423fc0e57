"""Parity at sizes beyond BASELINE.json's largest: multigrid clone vs the C restatement (exact eigenvalues, all host
cores).  Usage: python tools/large_roi_check.py 8192x8192 12000x7000 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc

inst = capi.Instance(0)
nt = min(oc.max_threads(), len(os.sched_getaffinity(0)), 16)
for spec in sys.argv[1:] or ["8192x8192"]:
    W, H = (int(v) for v in spec.split("x"))
    t = time.time(); dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=64); t_gen = time.time() - t
    body = dst.copy()
    t = time.time(); rc = inst.run(patch, body, mask, cx, cy, allow_not_converged=True); t_gpu = time.time() - t
    i = inst.info()
    print(f"{spec}: rc {rc} cycles {i.sweeps} last_update {i.last_update:.4f} device {i.ms_device_total:.2f} ms "
          f"({W * H / i.ms_device_total / 1e3:.0f} Mpix/s) host call {t_gpu * 1e3:.1f} ms arena {i.device_bytes / 2**30:.2f} GiB "
          f"(inputs generated in {t_gen:.0f} s)", flush=True)
    # beyond ~12 870 pixels per side the reference's float tables are singular (2 cos(pi/(n+1)) rounds to 2.0f: its lowest mode divides by
    # zero and the float-table port returns NaN); the library then returns the exact system's solution: checked against the exact-denominator port
    singular = min(W, H) - 2 >= 12870
    t = time.time(); want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=nt, exact_den=singular); t_cpu = time.time() - t
    s = compare.image_diff_stats(want, body)
    print(f"   C oracle on {nt} threads {t_cpu:.1f} s;  {compare.format_stats(s)}", flush=True)
    del dst, patch, mask, body, want
