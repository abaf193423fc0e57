"""Times the single-sweep Jacobi kernel variants (SC_JT_TH = LDS tile height, SC_JROLL = register-rolling segment)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os; sys.path.insert(0, %r)
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "2048x2048,4096x4096,1000x1000").split(",")]
for (W, H) in shapes:
    rng = np.random.default_rng(1)
    U = rng.normal(100, 30, (3, H, W)).astype(np.float32); F = rng.normal(0, 10, (3, H, W)).astype(np.float32)
    inst.field_load(U, F)
    ms = min(inst.field_time_sweeps(0, 100, 1, 1.0) for _ in range(3))
    print("%%dx%%d" %% (W, H), "%%.1f us  %%.0f GB/s" %% (ms*1e3, 12.0*(W-2)*(H-2)*3/ms/1e6), flush=True)
''' % ROOT
for var in sys.argv[1:] or ["SC_JT_TH=16"]:
    print(var, flush=True)
    k, v = var.split("=")
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{k: v}))
