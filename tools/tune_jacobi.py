import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
for roi in (2048, 4096):
    rng = np.random.default_rng(1)
    U = rng.normal(100, 30, (3, roi, roi)).astype(np.float32); F = rng.normal(0, 10, (3, roi, roi)).astype(np.float32)
    inst.field_load(U, F)
    ms = min(inst.field_time_sweeps(0, 100, 1, 1.0) for _ in range(3))
    print(roi, "%%.1f us  %%.0f GB/s" %% (ms*1e3, 12.0*(roi-2)**2*3/ms/1e6), flush=True)
''' % ROOT
for th in ("16", "32", "64"):
    print("SC_JT_TH=" + th, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SC_JT_TH=th))
