"""Diagnostic: clones, same-size groups, size classes and the mixed batch with SC_FLAG_POISON_ARENA (every block the arena hands out
without zeroing is filled with NaN bytes first) against the same calls without it: the bytes must be the same -- a difference is a
read of device memory nobody wrote.  python tests/tools/poison_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seamlesscloneoptimization_amd import capi
from oracle import oracle_np as o


def batch(sizes, flags, method=None, seed=0):
    items = [o.synth_inputs(W, H, seed_dst=40 + k + seed, seed_patch=90 + k + seed, margin=36) for k, (W, H) in enumerate(sizes)]
    inst = capi.Instance(0)
    kw = {"flags": flags}
    if method is not None:
        kw["method"] = method
    inst.set_solver(**kw)
    jobs = capi.Pool.make_jobs(len(items)); keep = []
    for j, (dst, patch, mask, cx, cy) in zip(jobs, items):
        f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(np.zeros_like(dst)), inst.to_device(mask)
        keep.append((f, b0, b, m, dst.shape))
        j.face, j.face_cols, j.face_rows, j.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
        j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
        j.centerX, j.centerY, j.body_restore = cx, cy, b0
    outs = []
    for rep in range(2):
        inst.run_device_batch(jobs)
        outs.append([inst.from_device(b, shape) for (f, b0, b, m, shape) in keep])
    inst.destroy()
    return outs


def solo(W, H, flags, method):
    dst, patch, mask, cx, cy = o.synth_inputs(W, H, seed_dst=7, seed_patch=8, margin=36)
    inst = capi.Instance(0); inst.set_solver(flags=flags, method=method)
    outs = []
    for rep in range(2):
        body = dst.copy(); inst.run(patch, body, mask, cx, cy); outs.append([body])
    inst.destroy()
    return outs


P = capi.SC_FLAG_POISON_ARENA
cases = [("mixed batch", lambda f: batch([(300, 310), (1003, 1010), (318, 333), (640, 480), (1020, 1001), (640, 480), (90, 70), (340, 305), (1012, 1024)], f)),
         ("class 1000s then 300s (buffers shrink)", lambda f: batch([(1003, 1010), (1020, 1001), (1012, 1024), (300, 310), (318, 333), (340, 305)], f)),
         ("class 150s", lambda f: batch([(154, 160), (150, 171), (165, 158), (158, 164), (161, 152)], f)),
         ("class 2100s", lambda f: batch([(2040, 2100), (2140, 2120), (2085, 2170)], f)),
         ("same-size 640x480 x4", lambda f: batch([(640, 480)] * 4, f)),
         ("same-size 157x157 x5", lambda f: batch([(157, 157)] * 5, f)),
         ("float fields class", lambda f: batch([(300, 310), (318, 333), (340, 305)], f | capi.SC_FLAG_FLOAT_FIELD | capi.SC_FLAG_FLOAT_RHS)),
         ("solo mg 2048", lambda f: solo(2048, 2048, f, capi.SC_METHOD_MULTIGRID)),
         ("solo mg 700x333", lambda f: solo(700, 333, f, capi.SC_METHOD_MULTIGRID)),
         ("solo fft 592", lambda f: solo(592, 592, f, capi.SC_METHOD_FFT)),
         ("solo dst 300x194", lambda f: solo(300, 194, f, capi.SC_METHOD_DST)),
         ("solo auto 154x100", lambda f: solo(154, 100, f, capi.SC_METHOD_AUTO))]
bad = 0
for name, fn in cases:
    a, b = fn(0), fn(P)
    for rep in range(2):
        for k, (x, y) in enumerate(zip(a[rep], b[rep])):
            if not np.array_equal(x, y):
                d = np.abs(x.astype(np.int16) - y.astype(np.int16)).max(axis=2); ys, xs = np.nonzero(d); bad += 1
                print(name, "rep", rep, "member", k, "differs in", len(ys), "pixels, max", int(d.max()), "box x", int(xs.min()), int(xs.max()), "y", int(ys.min()), int(ys.max()), x.shape, flush=True)
    print(name, "done", flush=True)
print("differing members", bad)
