"""Times the single-sweep Jacobi kernels: the register-rolling default (rows = 0) against the LDS-tiled 256 x rows kernel
(sc_solver_opts.jacobi_tile_rows = 16 | 32 | 64).  python tools/tune_jacobi.py [rows ...]   SHAPES=WxH,... in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
inst = capi.Instance(0)
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "2048x2048,4096x4096,1000x1000").split(",")]
for rows in [int(a) for a in sys.argv[1:]] or [0, 16, 32, 64]:
    inst.set_solver(jacobi_tile_rows=rows)
    for (W, H) in shapes:
        rng = np.random.default_rng(1)
        U = rng.normal(100, 30, (3, H, W)).astype(np.float32); F = rng.normal(0, 10, (3, H, W)).astype(np.float32)
        inst.field_load(U, F)
        ms = min(inst.field_time_sweeps(0, 100, 1, 1.0) for _ in range(3))
        print("rows %2d  %dx%d  %.1f us  %.0f GB/s" % (rows, W, H, ms * 1e3, 12.0 * (W - 2) * (H - 2) * 3 / ms / 1e6), flush=True)
