"""GPU box: k_mg_tail against the three launches it replaces (SC_FLAG_SEPARATE_TAIL) and against the numpy spec, one cycle, at given sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import mg_np
hip = capi.Instance(0)
for a in sys.argv[1:]:
    W, H = (int(v) for v in a.split("x"))
    rng = np.random.default_rng(W * 3 + H)
    U = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    F = np.zeros((3, H, W), np.float32)
    F[:, 1:-1, 1:-1] = rng.normal(0, 30, (3, H - 2, W - 2)).astype(np.float32)
    got = {}
    for flags in (0, capi.SC_FLAG_LEGACY_PATHS):
        hip.set_solver(method=capi.SC_METHOD_MULTIGRID, flags=flags | capi.SC_FLAG_KEEP_FIELD, legacy_paths=capi.SC_LEGACY_SEPARATE_TAIL, max_sweeps=1, update_tol=1e-30)
        hip.field_load(U, F)
        hip.field_solve(allow_not_converged=True)
        got[flags] = hip.field_store()
    lv = mg_np.build_levels(W, H)
    d = mg_np.direct_level(lv)
    spec = mg_np.solve(U[0], F[0], cycles=1) if W * H < 600 * 600 else None
    print(a, "levels", [(x.n, y.n) for x, y in lv[1:(d or 0) + 1]], "tail-vs-separate", float(np.abs(got[0] - got[capi.SC_FLAG_LEGACY_PATHS]).max()),
          "vs spec", None if spec is None else (float(np.abs(got[0][0] - spec).max()), float(np.abs(got[capi.SC_FLAG_LEGACY_PATHS][0] - spec).max())), flush=True)
