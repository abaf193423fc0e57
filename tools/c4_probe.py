"""BASELINE config 4 under the profiler: a 4096^2 ROI (604 MB of fields: the one size that streams from HBM instead of the
Infinity Cache) -- one full clone, then isolated launches of the single-sweep Jacobi kernels: rows rolling through registers
(k_jacobi_roll<4>, default) and the LDS-staged tiles with halo (k_jacobi<16>, k_jacobi<32>: the form the north-star names).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/c4_probe.py [launches]        (and again with WRITE_SIZE)
Prints one JSON line with the hipEvent timings of the same launches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
roi = 4096
inst = capi.Instance(0)
dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=64)
d_f, d_b, d_b0, d_m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
out = {"roi": roi, "launches": launches}
for rep in range(2):                                 # the second call is the steady state
    inst.copy_d2d_async(d_b, d_b0, dst.nbytes)
    inst.run_device(d_f, patch.shape[:2], d_b, dst.shape[:2], d_m, mask.shape[:2], cx, cy, sync=True)
i = inst.info()
out["full_clone"] = {"device_ms": round(i.ms_device_total, 4), "cycles": i.sweeps, "method": i.method}
inst.build_rhs(patch, dst, mask, cx, cy)             # float fields for the sweep kernels
alg = 12.0 * (roi - 2) * (roi - 2) * 3
for key, rows in (("k_jacobi_roll<4>", 0), ("k_jacobi<16>", 16), ("k_jacobi<32>", 32)):
    inst.set_solver(jacobi_tile_rows=rows)
    ms = inst.field_time_sweeps(capi.SC_METHOD_JACOBI, launches, 1, 1.0)
    out[key] = {"us_per_launch": round(ms * 1e3, 2), "algorithmic_GBps": round(alg / (ms * 1e-3) / 1e9, 1)}
inst.set_solver(jacobi_tile_rows=0)
for p in (d_f, d_b, d_b0, d_m):
    inst.free(p)
inst.destroy()
print(json.dumps(out))
