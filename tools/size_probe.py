"""GPU box: steady device time of one clone at given ROI sizes (WxH ...), and which level the bottom solves directly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from seamlesscloneoptimization_amd import capi
from oracle import mg_np
import _synth as o
inst = capi.Instance(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--fft" in sys.argv:                                # the default's direct solve (double transforms), forced
    inst.set_solver(flags=capi.SC_FLAG_NO_STAGE_MARKS | capi.SC_FLAG_FFT_FP64, method=capi.SC_METHOD_FFT)
else:
    inst.set_solver(flags=capi.SC_FLAG_NO_STAGE_MARKS, **({"method": capi.SC_METHOD_MULTIGRID} if "--mg" in sys.argv else {}))
if False:                                             # (round 4 experiments used sc_solver_opts.reserved[0]; the field is legacy_paths now)
    import ctypes as C_
    o_ = inst.get_solver(); o_.reserved[0] = 1
    assert inst.L.sc_hip_set_solver(inst.h, C_.byref(o_)) == 0
for a in args:
    W, H = (int(v) for v in a.split("x"))
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=64)
    d = [inst.to_device(x) for x in (patch, dst, mask, dst)]
    t = []
    for i in range(12):
        inst.copy_d2d_async(d[1], d[3], dst.nbytes); inst.sync()
        inst.run_device(d[0], patch.shape, d[1], dst.shape, d[2], mask.shape, cx, cy, sync=True)
        t.append(inst.info().ms_device_total)
    t = sorted(t[2:])
    lv = mg_np.build_levels(W, H)
    b, dl = mg_np.bottom_start(lv), mg_np.direct_level(lv)
    print(f"{W}x{H}: median {t[len(t)//2]:.4f} ms min {t[0]:.4f} cycles {inst.info().sweeps}  bottom_start {b} {(lv[b][0].n, lv[b][1].n)} direct {dl} {(lv[dl][0].n, lv[dl][1].n) if dl is not None else None}  Mpix/s {W*H/t[len(t)//2]/1e3:.0f}", flush=True)
    for p in d: inst.free(p)
