"""One clone at a time (no batching) for a kernel-trace timeline: python tools/solo_trace.py [roi] [n] [mg_direct_max] [flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o
roi = sys.argv[1] if len(sys.argv) > 1 else '2048'
roi_w, roi_h = (int(v) for v in roi.split('x')) if 'x' in roi else (int(roi), int(roi))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
inst = capi.Instance(0)
extra = {}
if len(sys.argv) > 3:
    extra["mg_direct_max"] = int(sys.argv[3])          # bottom kernel: largest level solved directly
if len(sys.argv) > 4:
    extra["flags"] = int(sys.argv[4])
if extra:
    inst.set_solver(**extra)
if len(sys.argv) > 5:                                   # experiments: sc_solver_opts.reserved[0]
    o_ = inst.get_solver(); o_.reserved[0] = int(sys.argv[5])
    import ctypes as C_
    assert inst.L.sc_hip_set_solver(inst.h, C_.byref(o_)) == 0
dst, patch, mask, cx, cy = o.synth_inputs(roi_w, roi_h, margin=256)
d_face, d_body, d_mask, d_keep = (inst.to_device(a) for a in (patch, dst, mask, dst))
tot = []
for i in range(n):
    inst.copy_d2d_async(d_body, d_keep, dst.nbytes)
    inst.sync()
    inst.run_device(d_face, patch.shape, d_body, dst.shape, d_mask, mask.shape, cx, cy)
    inst.sync()
    tot.append(inst.info().ms_device_total)
i = inst.info()
tot = sorted(tot[2:] or tot)
print("device_total over %d clones: min %.4f median %.4f max %.4f ms" % (len(tot), tot[0], tot[len(tot) // 2], tot[-1]))
print("device_total %.3f ms  mask %.3f pre %.3f solve %.3f post %.3f cycles %d" % (i.ms_device_total, i.ms_mask, i.ms_pre, i.ms_solve, i.ms_post, i.sweeps))
