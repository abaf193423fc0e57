"""What bounds the drop-in host call: raw PCIe copies from / to pinned memory against the call's own h2d / d2h stages.
python tools/host_path_probe.py [roi]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as o

roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
inst = capi.Instance(0)
for mb in (4.2, 12.6, 25.2):
    n = int(mb * 1e6)
    h, hnd = inst.pinned_array((n,))
    h[:] = 7
    d = inst.malloc(n)
    pg = np.full(n, 3, np.uint8)
    for name, fn in (("H2D pinned", lambda: inst.L.sc_hip_memcpy_h2d(inst.h, d, h.ctypes.data, n)),
                     ("D2H pinned", lambda: inst.L.sc_hip_memcpy_d2h(inst.h, h.ctypes.data, d, n)),
                     ("H2D pageable", lambda: inst.L.sc_hip_memcpy_h2d(inst.h, d, pg.ctypes.data, n)),
                     ("host memcpy (numpy, 1 thread)", lambda: np.copyto(h, pg))):
        fn(); fn()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{mb:5.1f} MB {name:30s} median {ts[7] * 1e3:.3f} ms = {n / ts[7] / 1e9:.1f} GB/s  (min {ts[0] * 1e3:.3f})", flush=True)
    inst.free(d); inst.free_pinned(hnd)
dst, patch, mask, cx, cy = o.synth_inputs(roi, roi, margin=256)
body = dst.copy()
inst.run(patch, body, mask, cx, cy); inst.run(patch, body, mask, cx, cy)
for rep in range(2):
    for sched in (0,):
        rows = []
        for _ in range(20):
            body[...] = dst
            t0 = time.perf_counter(); inst.run(patch, body, mask, cx, cy); t = (time.perf_counter() - t0) * 1e3
            i = inst.info()
            rows.append((t, i.ms_h2d, i.ms_device_total, i.ms_d2h, i.ms_call))
        rows.sort()
        m = rows[len(rows) // 2]
        print("host call %dx%d schedule %d: median call %.3f ms (h2d %.3f, device %.3f, d2h %.3f, stream %.3f); min %.3f max %.3f" % (roi, roi, sched, m[0], m[1], m[2], m[3], m[4], rows[0][0], rows[-1][0]))
# the same images page-locked by the caller at the library's device pitch: no packing at all
