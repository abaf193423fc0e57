"""Isolated launches of the hot kernels for counter collection: rocprofv3 --pmc ... -- python3 tools/pmc_probe.py [roi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o
roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
inst = capi.Instance(0)
dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=64)
body = dst.copy(); inst.run(patch, body, mask, cx, cy)
print("cycle0", inst.time_cycle0(6))
inst.build_rhs(patch, dst, mask, cx, cy)      # float fields for the sweep kernels (the clone left a float16 right-hand side)
print("rb4", inst.field_time_sweeps(1, 24, 4, 1.0))
print("jac8", inst.field_time_sweeps(0, 48, 8, 1.0))
print("jac1", inst.field_time_sweeps(0, 6, 1, 1.0))
