"""Scratch GPU validation: every kernel vs the oracle (superseded by tests/ -m gpu)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seamlesscloneoptimization_amd as pkg
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o, oracle_c as oc

inst = capi.Instance(0)
ok = True
def check(name, cond):
    global ok
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok = ok and bool(cond)

for (W, H, ell) in [(16, 12, False), (33, 17, False), (298, 192, False), (300, 260, True), (513, 129, False)]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=64, ellipse=ell)
    geo_c, M_c = oc.mask_stage(mask, cx, cy)
    geo, M = inst.mask_stage(mask, cx, cy)
    check(f"mask_stage {W}x{H} ell={ell} geo", np.array_equal(geo, geo_c))
    check(f"mask_stage {W}x{H} M", np.array_equal(M, M_c))
    B_c, lap_c = oc.build_rhs(dst, patch, geo_c, M_c)
    g2, B, lap = inst.build_rhs(patch, dst, mask, cx, cy)
    check(f"build_rhs {W}x{H} B", np.array_equal(B, B_c))
    check(f"build_rhs {W}x{H} lap", np.array_equal(lap, lap_c))
    for method, name, om in [(0, "jacobi", 1.0), (1, "rbgs", 1.0), (2, "sor", 1.6)]:
        inst.field_load(B_c, lap_c)
        inst.field_sweep(method, 5, om, 1)
        got = inst.field_store()
        want = oc.jacobi(B_c, lap_c, 5) if method == 0 else oc.rbgs(B_c, lap_c, 5, om if method == 2 else 1.0)
        check(f"{name} {W}x{H} bit-exact", np.array_equal(got, want))
        r = inst.field_residual(); rc = oc.residual(want, lap_c)
        check(f"residual {name} {W}x{H}", abs(r[0]-rc[0]) <= 1e-9*rc[0] and abs(r[1]-rc[1]) <= 1e-9*rc[1])

# full clone, c1-like
from PIL import Image
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sky = np.ascontiguousarray(np.asarray(Image.open(G + "/sky.jpg"))[:, :, ::-1])
air = np.ascontiguousarray(np.asarray(Image.open(G + "/airplane.jpg"))[:, :, ::-1])
mask = np.full(air.shape[:2], 255, np.uint8)
want = o.seamless_clone(sky, air, mask, 800, 150)
for method, tol in [(3, 0.0), (2, 2e-5)]:
    body = sky.copy()
    inst.set_solver(method=method, tol=tol, max_sweeps=(30 if method == 3 else 100000), check_every=64)
    t = time.time(); rc = inst.run(air, body, mask, 800, 150, sync=True); dt = time.time() - t
    i = inst.info()
    s = pkg.compare.image_diff_stats(want, body)
    print("c1 method", method, "rc", rc, "sweeps", i.sweeps, "rel", i.rel_residual, "ms solve", i.ms_solve, "total", i.ms_device_total, "wall", dt*1e3, pkg.compare.format_stats(s))
    check(f"c1 clone method {method} within 1", s["max"] <= 1)

inst.set_solver(method=3, tol=0.0, max_sweeps=30)
for (W, H) in [(77, 53), (1000, 39), (1026, 770), (2048, 2048)]:
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, margin=64)
    want = o.seamless_clone(dst, patch, mask, cx, cy)
    body = dst.copy()
    rc = inst.run(patch, body, mask, cx, cy, sync=True, allow_not_converged=True)
    i = inst.info(); s = pkg.compare.image_diff_stats(want, body)
    print(f"MG {W}x{H} rc {rc} cycles {i.sweeps} ms: mask {i.ms_mask:.3f} pre {i.ms_pre:.3f} solve {i.ms_solve:.3f} post {i.ms_post:.3f} h2d {i.ms_h2d:.3f} d2h {i.ms_d2h:.3f}", pkg.compare.format_stats(s), flush=True)
    check(f"MG clone {W}x{H} within 1", s["max"] <= 1 and rc == 0)
# timing of sweep kernels
for (W, H) in [(2048, 2048), (4096, 4096)]:
    rng = np.random.default_rng(1)
    U = rng.normal(100, 30, (3, H, W)).astype(np.float32); F = rng.normal(0, 10, (3, H, W)).astype(np.float32)
    inst.field_load(U, F)
    unknowns = (W-2)*(H-2)*3
    for method, name, spls in [(0, "jacobi", (1, -1, 2, 3, 4)), (1, "rbgs", (1, -1, 2))]:
        for spl in spls:
            ms = inst.field_time_sweeps(method, 50, spl, 1.0)
            sweeps_per_launch = 0.5 if (method == 1 and spl == 1) else abs(spl)
            by = 12 * unknowns * sweeps_per_launch
            print(f"{name} spl={spl} {W}x{H}: {ms*1e3:.1f} us/launch -> {by/ms/1e9:.2f} TB/s algorithmic", flush=True)
inst.destroy()
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
