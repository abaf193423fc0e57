"""Cycles, last correction and parity against the float-table port for V(pre, post) schedules (sc_solver_opts.mg_pre / mg_post), with and
without the composed level 1: python tests/tools/v_schedule_probe.py   """
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc
oc.build()
inst = capi.Instance(0)
for (W, H) in [(2048, 2048), (1030, 1000), (4096, 4096)]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=24)
    want = oc.seamless_clone(dst, patch, mask, cx, cy, nthreads=16, exact_den=False)
    for (pre, post) in [(2, 2), (1, 2), (2, 1), (1, 1)]:
        for flags in (0, capi.SC_FLAG_NO_COMPOSE_L1):
            inst.set_solver(mg_pre=pre, mg_post=post, flags=flags)
            body = dst.copy()
            rc = inst.run(patch, body, mask, cx, cy, allow_not_converged=True)
            i = inst.info()
            s = compare.image_diff_stats(want, body)
            print(W, H, "V(%d,%d)" % (pre, post), "flags", flags, "rc", rc, "cycles", i.sweeps, "last %.4f" % i.last_update, "device %.3f ms" % i.ms_device_total, "max", s["max"], "pct %.4f" % s["percent"], flush=True)
