"""GPU box: where does the SECOND host-image call of an instance spend what it spends over the steady call (bench.py: pcie.first_two_calls_ms)?
Wall time of calls 1..6 of a fresh instance at 2048^2, five ways:
  pageable_restored   numpy images, destination restored by a host copy before every call (what bench.py's leg does)
  pageable_untouched  numpy images, destination left as the previous call wrote it
  pageable_new_pages  numpy images, a FRESH destination array (pages never seen by the runtime) for every call
  pinned              images in sc_hip_host_alloc'ed memory (restored by a host copy)
  pageable_twice      one instance after another in the same process (does the process, not the instance, pay?)
python tools/second_call_probe.py [roi]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seamlesscloneoptimization_amd import capi

roi = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(1)
Wd = Hd = roi + 64
dst = np.clip(128.0 + rng.normal(0.0, 14.0, (Hd, Wd, 3)), 0, 255).astype(np.uint8)
patch = rng.integers(0, 256, (roi + 2, roi + 2, 3), dtype=np.uint8)
mask = np.full((roi + 2, roi + 2), 255, np.uint8)
cx, cy = Wd // 2, Hd // 2


def calls(inst, get_body, p, m, n=6):
    out = []
    for k in range(n):
        body = get_body(k)
        t0 = time.perf_counter()
        inst.run(p, body, m, cx, cy)
        out.append(round((time.perf_counter() - t0) * 1e3, 3))
    i = inst.info()
    out.append({"last_call_stream_ms": round(i.ms_call, 3), "h2d": round(i.ms_h2d, 3), "device": round(i.ms_device_total, 3), "d2h": round(i.ms_d2h, 3)})
    return out


res = {"roi": roi}
warm = capi.Instance(0)                       # the process's first instance takes the runtime's own start-up: not what is asked here
warm.run(patch, dst.copy(), mask, cx, cy)
warm.destroy()

inst = capi.Instance(0); inst.set_solver(flags=0)
body = dst.copy()
def restored(k):
    body[...] = dst
    return body
res["pageable_restored"] = calls(inst, restored, patch, mask)
inst.destroy()

inst = capi.Instance(0); inst.set_solver(flags=0)
body2 = dst.copy()
res["pageable_untouched"] = calls(inst, lambda k: body2, patch, mask)
inst.destroy()

inst = capi.Instance(0); inst.set_solver(flags=0)
res["pageable_new_pages"] = calls(inst, lambda k: dst.copy(), patch, mask)
inst.destroy()

inst = capi.Instance(0); inst.set_solver(flags=0)
pb, hb = inst.pinned_array(dst.shape); pp, hp = inst.pinned_array(patch.shape); pm, hm = inst.pinned_array(mask.shape)
pp[...] = patch; pm[...] = mask
def pinned(k):
    pb[...] = dst
    return pb
res["pinned"] = calls(inst, pinned, pp, pm)
for h in (hb, hp, hm):
    inst.free_pinned(h)
inst.destroy()

inst = capi.Instance(0); inst.set_solver(flags=0)
res["pageable_twice"] = calls(inst, restored, patch, mask)
inst.destroy()
print(json.dumps(res))
