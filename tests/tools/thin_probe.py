"""Why thin ROIs show the largest share of channels off by one in tests/tools/fuzz_shapes.py (13.8 % of the image = ALL unknowns of a
5 x 518 ROI): three erodes empty a mask narrower than 7 pixels, the guidance field is then the destination's own gradient and the
exact solution is the destination itself -- integers, up to the reflect-101 edge terms.  Truncation of integer +- 1e-5 is a coin
flip for ANY arithmetic (the numpy port with float tables differs from the numpy port with exact tables on 4385 of 4644 channels while
their fields differ by 1e-4; the GPU field matches the C port's exact solve to 5e-5).  python tests/tools/thin_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from oracle import oracle_np as o, oracle_c as oc
oc.build()
inst = capi.Instance(0)
for (W, H, case) in [(5, 518, 382), (4, 300, 1), (7, 640, 2), (12, 400, 3), (20, 800, 4), (600, 5, 5), (900, 9, 6)]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=case, seed_patch=1000 + case, margin=16)
    want = o.seamless_clone(dst, patch, mask, cx, cy, float_tables=True)
    want_e = o.seamless_clone(dst, patch, mask, cx, cy, float_tables=False)
    for flags in (capi.SC_FLAG_KEEP_FIELD, capi.SC_FLAG_KEEP_FIELD | capi.SC_FLAG_EXACT_TABLES, 0):
        inst.set_solver(flags=flags)
        body = dst.copy(); inst.run(patch, body, mask, cx, cy)
        i = inst.info()
        s = compare.image_diff_stats(want, body); se = compare.image_diff_stats(want_e, body)
        line = "%dx%d flags %d cycles %d conv %d last %.4f | vs float port: max %d n %d | vs exact port: max %d n %d" % (W, H, flags, i.sweeps, i.converged, i.last_update, s["max"], s["diff_channels"], se["max"], se["diff_channels"])
        if flags & capi.SC_FLAG_KEEP_FIELD:
            U = inst.field_store()
            geo, M = oc.mask_stage(mask, cx, cy); B, lap = oc.build_rhs(dst, patch, geo, M); g = oc.fold(B, lap)
            ue = oc.solve_dst(g, 8, exact_den=True); uf = oc.solve_dst(g, 8, exact_den=False)
            line += " | field err vs exact %.4f, exact-vs-float tables %.4f" % (np.abs(U[:, 1:-1, 1:-1] - ue).max(), np.abs(ue - uf).max())
        print(line, flush=True)
