"""A few SC_METHOD_DST clones for counter collection: rocprofv3 --pmc ... -- python3 tools/dst_probe.py [roi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import _synth as o
from seamlesscloneoptimization_amd import capi
roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
inst = capi.Instance(0)
inst.set_solver(method=capi.SC_METHOD_DST)
dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=64)
for _ in range(3):
    body = dst.copy(); inst.run(patch, body, mask, cx, cy)
i = inst.info()
print("dst solve %.3f ms, device %.3f ms" % (i.ms_solve, i.ms_device_total))
