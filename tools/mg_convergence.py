"""Prints the per-cycle max coarse-grid correction (the multigrid stop-rule quantity) and the
max |delta| / %% differing vs a tightly converged clone, for a few inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from seamlesscloneoptimization_amd import capi, compare
from bench import synth
from PIL import Image
inst = capi.Instance(0)
G = os.path.join(ROOT, "tests", "golden")
sky = np.ascontiguousarray(np.asarray(Image.open(G + "/sky.jpg"))[:, :, ::-1]); air = np.ascontiguousarray(np.asarray(Image.open(G + "/airplane.jpg"))[:, :, ::-1])
cases = {"c1": (sky, air, np.full(air.shape[:2], 255, np.uint8), 800, 150)}
for roi in (512, 1024, 2048):
    cases[f"synth{roi}"] = synth(roi, 0)
for name, (dst, patch, mask, cx, cy) in cases.items():
    inst.set_solver(method=3, max_sweeps=12, update_tol=1e-6, tol=0.0)
    ref = dst.copy(); inst.run(patch, ref, mask, cx, cy, allow_not_converged=True)
    line = []
    for k in range(3, 8):
        inst.set_solver(method=3, max_sweeps=k, update_tol=1e-30, tol=0.0)
        b = dst.copy(); inst.run(patch, b, mask, cx, cy, allow_not_converged=True)
        s = compare.image_diff_stats(ref, b)
        line.append("k=%d corr=%.4f max=%d diff=%.4f%%" % (k, inst.info().last_update, s["max"], s["percent"]))
    print(name, " | ".join(line), flush=True)
