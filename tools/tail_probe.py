"""GPU box: phases of the k_mg_tail launch (shader-clock cycles between its boundaries) and the coarse chain with / without it."""
import sys; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from seamlesscloneoptimization_amd import capi
import _synth as o
names = ["load F", "pre-smooth", "residual+restrict", "product 1", "product 2", "product 3", "product 4", "prolong+exchange", "post-smooth", "stores"]
inst = capi.Instance(0)
for roi in [int(a) for a in sys.argv[1:]] or (2048, 1022):
    dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=64)
    d = [inst.to_device(a) for a in (patch, dst, mask)]
    for flags in (0, capi.SC_FLAG_LEGACY_PATHS):
        inst.set_solver(method=capi.SC_METHOD_MULTIGRID, flags=flags, legacy_paths=capi.SC_LEGACY_SEPARATE_TAIL)
        inst.run_device(d[0], patch.shape, d[1], dst.shape, d[2], mask.shape, cx, cy)
        print(roi, "separate" if flags else "tail", [tuple(round(v, 5) if isinstance(v, float) else v for v in inst.time_coarse_chain(50)) for _ in range(3)])
        if not flags:
            for rep in range(3):
                ph = inst.time_tail_phases()
                print("   ", sum(ph), " ".join(f"{n}={v}" for n, v in zip(names, ph)))
    for p in d: inst.free(p)
