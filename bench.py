#!/usr/bin/env python3
"""bench.py -- seamless-clone throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns one rank per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks from the environment)

A "step" is one pass of the hot path over one batch: `--batch` (default 32) independent
NORMAL_CLONEs of a 2048x2048 ROI per GPU (mask stage + fused pre-process + Poisson solve + float-table
correction + fused post-process; result = the reference's arithmetic, +-1 grey level), through the C ABI with
their images already resident in HBM, issued by the library's native pool (sc_hip_pool_run: `--streams`,
default 2, instances = HIP streams, one C++ worker thread each).  Each worker takes `--group` (default 16)
clones at a time and solves them as ONE field of 3n channels (sc_hip_run_device_batch: same-size ROIs share
one set of solver launches); `--group 1` is one clone per set of launches (sc_hip_run_device).  Every rank
owns its own synthetic images (weak scaling: independent images, no data-path collective);
value = total ROI Mpix / max-over-ranks wall time.  Each destination is restored from a pristine
device copy before every clone (inside the timed region) so no clone starts from an already-converged field.

The same JSON line carries `value_without_in_step_restore` (the step minus its destination refresh), `single_clone` (ONE 2048^2 clone alone: BASELINE config 3 as written), `value_float32_storage` (the
same timed step with float32 fields throughout), `roofline` (the dominant kernel AS THE TIMED REGION RUNS IT -- the grouped
level-0 multigrid launch -- HIP-event timed on the library's stream; plus all four level-0 launch forms of a solve), `pcie` (the
drop-in host-image call: median / p95 / min of 24), `new_size` (the first call at a ROI size the instance has never seen),
`per_rank` (every rank's own elapsed time and device) and `cpu_baseline` (the C restatement of what cv::seamlessClone
computes, timed on the host cores; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--roi", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=32, help="independent images per GPU per step")
    ap.add_argument("--streams", type=int, default=2, help="concurrent library instances (HIP streams) per GPU")
    ap.add_argument("--group", type=int, default=None, help="clones of one ROI size a worker solves as ONE field of 3n channels (sc_hip_run_device_batch); default 16, --config c5: the rank's images split evenly over its streams, at most 32 per group (64 x 1024^2 on one GPU: 2 x 32 = 19.5 Gpix/s, 2 x 16 = 18.3)")
    ap.add_argument("--method", default="mg", choices=["jacobi", "mg", "rbgs", "sor", "auto", "fft", "dst"],
                    help="solver of the timed region (default mg: what SC_METHOD_AUTO resolves to at the 2048^2 ROI of the metric)")
    ap.add_argument("--sweeps-per-launch", type=int, default=0)
    ap.add_argument("--exact-tables", action="store_true", help="time the exact 5-point solution (SC_FLAG_EXACT_TABLES) instead of the reference's float-table answer")
    ap.add_argument("--extra-flags", type=int, default=0, help="further sc_solver_opts.flags bits (A/B runs of a variant, e.g. 2048 = SC_FLAG_FLOAT_FIELD)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--kernel-launches", type=int, default=100, help="launches in the roofline micro-region")
    ap.add_argument("--config", default="c3", choices=["c3", "c5"],
                    help="c3 (default): the metric's configuration -- 2048^2 ROIs, --batch clones per GPU per step, weak scaling.  "
                         "c5: BASELINE config 5 as written -- 64 independent 1024^2 clones in total, image i on rank i mod N "
                         "(8 per GPU at N = 8), strong scaling")
    ap.add_argument("--affinity", default="auto", choices=["auto", "none"],
                    help="N > 1: pin each rank (and the library threads it starts) to its share of the host cores (auto) or leave the mask alone")
    ap.add_argument("--host-calls", type=int, default=24, help="timed drop-in host-image calls of the `pcie` leg (median / p95 / min are reported)")
    ap.add_argument("--no-float32-leg", action="store_true", help="skip value_float32_storage (the same timed step with float32 fields and right-hand side)")
    ap.add_argument("--no-fresh-leg", action="store_true", help="skip value_without_in_step_restore (up to 16 steps, each into its own pre-resident destinations: 0.43 GB of HBM per step)")
    ap.add_argument("--no-c5-projection", action="store_true", help="skip the c5_projected_at_8 leg (config 5's 8-image shard on this GPU, x 8)")
    ap.add_argument("--no-mixed-sizes", action="store_true", help="skip the mixed_sizes leg (64 clones of 64 different ROI sizes through the pool)")
    ap.add_argument("--no-new-size", action="store_true", help="skip the new_size leg (first call at a ROI size the instance has not seen)")
    ap.add_argument("--no-c4", action="store_true", help="skip the roofline_c4 leg (config 4: single-sweep Jacobi kernels at a 4096^2 ROI, HBM bound)")
    ap.add_argument("--reference-table", action="store_true",
                    help="instead of the flagship line: the reference's own published table (PDF p3) -- end-to-end latency of the drop-in "
                         "call at source patches 154x100, 300x194, 592x592, 2400x1552, its protocol (1 warm-up + 50 rounds), the CPU port beside it")
    return ap.parse_args(argv)


def synth(roi, rank):
    """SURVEY 8d synthetic inputs (numpy only; the oracle module's generator restated so the
    product bench does not import oracle/ for its GPU leg)."""
    import numpy as np
    W = H = roi
    margin = 256
    Hd, Wd = H + margin, W + margin
    rng = np.random.default_rng(1001 + rank)
    yy, xx = np.mgrid[0:Hd, 0:Wd]
    base = 128.0 + 60.0 * np.sin(2 * np.pi * xx / Wd) * np.cos(2 * np.pi * yy / Hd)
    dst = np.clip(base[:, :, None] + rng.normal(0.0, 12.0, (Hd, Wd, 3)), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(2002 + rank)
    Hp, Wp = H + 2, W + 2
    yy, xx = np.mgrid[0:Hp, 0:Wp]
    base = 110.0 + 50.0 * np.cos(3 * np.pi * xx / max(W, 1))
    patch = np.clip(base[:, :, None] + rng.normal(0.0, 20.0, (Hp, Wp, 3)), 0, 255).astype(np.uint8)
    mask = np.full((Hp, Wp), 255, np.uint8)
    return dst, patch, mask, Wd // 2, Hd // 2


class BatchSynth:
    """The same images for a whole batch without 25 s of numpy in front of 0.13 s of GPU work: the smooth parts and ONE pair of
    unit-variance noise fields are built once per rank (float32); image k is smooth + sigma x (the noise shifted by a k-dependent
    offset), clipped to 8 bits -- the SURVEY 8d statistics (same smooth parts, N(0, 12) / N(0, 20) pixel noise), every image
    different from every other, ~0.1 s each instead of ~0.8.  Image 0 of rank 0 is synth(roi, 0) itself, bit for bit: the CPU
    baseline and the parity figures of the JSON line are computed on it."""

    def __init__(self, roi, seed):
        import numpy as np
        self.np = np
        self.roi = roi
        W = H = roi
        Hd, Wd = H + 256, W + 256
        Hp, Wp = H + 2, W + 2
        yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
        self.base_d = (128.0 + 60.0 * np.sin(2 * np.pi * xx / Wd) * np.cos(2 * np.pi * yy / Hd)).astype(np.float32)[:, :, None]
        xx = np.arange(Wp, dtype=np.float32)[None, :, None]
        self.base_p = (110.0 + 50.0 * np.cos(3 * np.pi * xx / max(W, 1))).astype(np.float32)
        rng = np.random.default_rng(seed)
        self.nd = rng.standard_normal((Hd, Wd, 3), dtype=np.float32)
        self.npatch = rng.standard_normal((Hp, Wp, 3), dtype=np.float32)
        self.mask = np.full((Hp, Wp), 255, np.uint8)
        self.centre = (Wd // 2, Hd // 2)

    def image(self, k):
        np = self.np
        sd = (37 * k + 11, 101 * k + 5)
        dst = np.clip(self.base_d + 12.0 * np.roll(self.nd, sd, axis=(0, 1)), 0, 255).astype(np.uint8)
        patch = np.clip(self.base_p + 20.0 * np.roll(self.npatch, (53 * k + 3, 71 * k + 29), axis=(0, 1)), 0, 255).astype(np.uint8)
        return dst, patch, self.mask, self.centre[0], self.centre[1]


def pin_rank_to_its_cores(local_rank, world, mode="auto", gpu_pci_bus_id=None):
    """One rank per GPU shares a host with world - 1 others, each with pool workers and copy helpers (sc_pool.cpp,
    sc_hostcopy.cpp): pin this process -- and so every thread it creates later -- to its share of the cores BEFORE the library
    starts its threads.  auto: the cores the kernel lists as local to the GPU's PCI device (/sys/bus/pci/devices/<bdf>/
    local_cpulist) split evenly among the ranks that share them, else an even split of the current affinity mask; none: leave
    the mask alone.  Returns the list of cores (what the JSON line reports)."""
    from seamlesscloneoptimization_amd.batch import rank_cores
    if mode == "none" or world <= 1 or not hasattr(os, "sched_setaffinity"):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    local = None
    if gpu_pci_bus_id:
        try:
            local = open(f"/sys/bus/pci/devices/{gpu_pci_bus_id.lower()}/local_cpulist").read().strip()
        except OSError:
            local = None
    cores = rank_cores(local_rank, world, sorted(os.sched_getaffinity(0)), local)
    if cores:
        os.sched_setaffinity(0, cores)
    return cores


def cpu_baseline(dst, patch, mask, cx, cy, gpu_out, gpu_out_exact, budget_s):
    """OpenCV-equivalent CPU restatement (oracle/sc_oracle.c, float32 eigenvalue tables exactly as the reference builds
    them), single thread (OpenCV's DFT path is serial), 1 warm-up + timed repeats inside `budget_s`.  The parity numbers
    compare like with like: the GPU's default output with the float-table port (the variant that is timed, = what the
    reference computes), the GPU's SC_FLAG_EXACT_TABLES output with the exact-denominator port."""
    from oracle import oracle_c as oc
    import numpy as np
    oc.build()
    W, H = mask.shape[1] - 2, mask.shape[0] - 2
    t0 = time.perf_counter()
    oc.seamless_clone(dst, patch, mask, cx, cy, 1, False)   # warm-up (also sizes the run)
    one = time.perf_counter() - t0
    reps = max(1, min(5, int(budget_s / max(one, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        oc.seamless_clone(dst, patch, mask, cx, cy, 1, False)
    dt = (time.perf_counter() - t0) / reps
    nthr = max(1, min(oc.max_threads(), len(os.sched_getaffinity(0)), 16))  # the GPU box gives 16 cores per GPU
    t0 = time.perf_counter()
    ref_f = oc.seamless_clone(dst, patch, mask, cx, cy, nthr, False)  # the reference's arithmetic, all cores
    dt_all = time.perf_counter() - t0
    ref_e = oc.seamless_clone(dst, patch, mask, cx, cy, nthr, True)   # exact denominators

    def diff(a, b):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        return {"maxdiff": int(d.max()), "percent_differing": round(float((d > 0).mean() * 100.0), 4)}
    # apples-to-apples for the fixed-iteration stencil config (512^2 ROI, 1000 Jacobi sweeps):
    # the same sweeps in C on one core, 200 timed and scaled to 1000
    rng = np.random.default_rng(5)
    Uj = rng.normal(100, 40, (3, 512, 512)).astype(np.float32)
    Fj = rng.normal(0, 20, (3, 512, 512)).astype(np.float32)
    t0 = time.perf_counter()
    oc.jacobi(Uj, Fj, 200)
    tj = (time.perf_counter() - t0) * 5.0
    return {
        "value": round(W * H / dt / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": f"{reps} timed + 1 warm-up full clones of the same {W}x{H} ROI, C restatement of "
                  f"cv::seamlessClone (DST-direct, float32 tables as the reference builds them), {dt * 1e3:.0f} ms each",
        "all_cores": {"value": round(W * H / dt_all / 1e6, 3), "cores": nthr, "note": "OpenMP over rows, 1 run"},
        "gpu_vs_float_table_port": dict(diff(gpu_out, ref_f), note="GPU default output vs the port that is timed above (float32 "
                                        "eigenvalue tables = the reference's / OpenCV's arithmetic; the port's FFT internals are double, "
                                        "real OpenCV runs a float32 DFT whose own rounding is unknown here)"),
        "gpu_exact_tables_vs_exact_den_port": dict(diff(gpu_out_exact, ref_e), note="GPU with SC_FLAG_EXACT_TABLES vs the port with "
                                                   "double denominators (the exact 5-point system)"),
        "float_table_port_vs_exact_den_port": dict(diff(ref_f, ref_e), note="how far the reference's own float tables are from the "
                                                   "exact system at this size"),
        "jacobi_512x512_1000_sweeps": {"ms": round(tj * 1e3, 1), "Gpix_updates_per_s": round(510 * 510 * 1000 / tj / 1e9, 3),
                                       "cores": 1, "note": "CPU counterpart of tools/bench_configs.py c2"},
    }


def reference_table(args):
    """The reference's published measurement (SeamlessClone Project Overview.pdf p3 / BASELINE.md section 1): end-to-end
    latency per clone -- host images in, result in the caller's image -- at its four source-patch sizes, "50 rounds with warm
    up".  One JSON line per size: the default path (multigrid + float-table correction), the direct DST path, and the CPU
    port (cpu_baseline leg: the only place bench.py touches oracle/) timed on the same inputs."""
    from seamlesscloneoptimization_amd import capi
    capi.load()
    import numpy as np
    from oracle import oracle_c as oc
    oc.build()
    published = {"154x100": {"V100_ms_fft_gemm": [1.651, 1.434], "T4_ms_fft_gemm": [2.185, 1.939]},
                 "300x194": {"V100_ms_fft_gemm": [1.968, 1.905], "T4_ms_fft_gemm": [2.911, 2.613]},
                 "592x592": {"V100_ms_fft_gemm": [5.401, 5.621], "T4_ms_fft_gemm": [9.047, 7.424]},
                 "2400x1552": {"V100_ms_fft_gemm": [63.988, 56.412], "T4_ms_fft_gemm": [91.306, 79.462]}}
    inst = capi.Instance(0)
    nthr = max(1, min(oc.max_threads(), len(os.sched_getaffinity(0)), 16))
    for pw, ph in ((154, 100), (300, 194), (592, 592), (2400, 1552)):
        # destination 1600 x 898 as in the reference (PDF p4) where the patch fits, else patch + 256
        dw, dh = (1600, 898) if pw + 2 <= 1600 and ph + 2 <= 898 else (pw + 256, ph + 256)
        rng = np.random.default_rng(pw * 7 + ph)
        yy, xx = np.mgrid[0:dh, 0:dw]
        dst = np.clip((128.0 + 60.0 * np.sin(2 * np.pi * xx / dw) * np.cos(2 * np.pi * yy / dh))[:, :, None] + rng.normal(0, 12, (dh, dw, 3)), 0, 255).astype(np.uint8)
        yy, xx = np.mgrid[0:ph, 0:pw]
        patch = np.clip((110.0 + 50.0 * np.cos(3 * np.pi * xx / pw))[:, :, None] + rng.normal(0, 20, (ph, pw, 3)), 0, 255).astype(np.uint8)
        mask = np.full((ph, pw), 255, np.uint8)
        cx, cy = dw // 2, dh // 2
        row = {"patch": f"{pw}x{ph}", "dst": [dw, dh], "protocol": "1 warm-up + 50 rounds, pageable host images, one application per call, "
               "destination restored on the host between rounds (restore time reported, not included)", "published": published[f"{pw}x{ph}"]}
        outs = {}
        for name, method, mflags in (("default_auto", capi.SC_METHOD_AUTO, 0), ("multigrid_plus_float_table_correction", capi.SC_METHOD_MULTIGRID, 0),
                                     ("direct_dst", capi.SC_METHOD_DST, 0), ("direct_fft", capi.SC_METHOD_FFT, 0),
                                     ("direct_fft_fp64", capi.SC_METHOD_FFT, capi.SC_FLAG_FFT_FP64)):
            if name == "direct_fft_fp64" and max(pw, ph) > 4096:
                continue
            inst.set_solver(method=method, flags=args.extra_flags | mflags)
            body = dst.copy()
            inst.run(patch, body, mask, cx, cy)
            t_run = t_restore = 0.0
            for _ in range(50):
                t0 = time.perf_counter(); body[...] = dst; t1 = time.perf_counter()
                inst.run(patch, body, mask, cx, cy)
                t2 = time.perf_counter()
                t_restore += t1 - t0; t_run += t2 - t1
            i = inst.info()
            outs[name] = body.copy()
            row[name] = {"ms_per_clone_end_to_end": round(t_run / 50 * 1e3, 4), "h2d_ms": round(i.ms_h2d, 4), "device_ms": round(i.ms_device_total, 4),
                         "d2h_ms": round(i.ms_d2h, 4), "restore_ms_not_included": round(t_restore / 50 * 1e3, 4), "roi": [i.W, i.H],
                         "device_bytes": int(i.device_bytes), "method_that_ran": {3: "multigrid", 4: "dst", 6: "fft"}.get(i.method, i.method) + (" (double transforms)" if name == "default_auto" and i.method == 6 else "")}
        inst.set_solver(method=capi.SC_METHOD_AUTO, flags=args.extra_flags)
        t0 = time.perf_counter(); ref = oc.seamless_clone(dst, patch, mask, cx, cy, 1, False); one = time.perf_counter() - t0
        reps = max(1, min(20, int(args.cpu_seconds / 4 / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(reps):
            oc.seamless_clone(dst, patch, mask, cx, cy, 1, False)
        row["cpu_port_1_core_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)

        def dstat(a, b):
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            return {"maxdiff": int(d.max()), "diff_sum": int(d.sum()), "percent_differing": round(float((d > 0).mean() * 100), 4)}
        for name, img in outs.items():
            row[name]["vs_float_table_port"] = dstat(img, ref)
        # How far is "+-1 against the port" from "+-1 against OpenCV"?  The port transforms in double; OpenCV's dft and cuFFT
        # transform in float32.  The same port with float32 transform internals (mixed radix; Bluestein) beside it:
        ref32 = oc.seamless_clone(dst, patch, mask, cx, cy, nthr, False, "f32")
        ref32b = oc.seamless_clone(dst, patch, mask, cx, cy, nthr, False, "f32_bluestein")
        row["float32_transform_bound"] = {
            "gpu_default_vs_f64_port": dstat(outs["default_auto"], ref), "gpu_default_vs_f32_port": dstat(outs["default_auto"], ref32),
            "f32_port_vs_f64_port": dstat(ref32, ref), "f32_bluestein_port_vs_f64_port": dstat(ref32b, ref),
            "f32_port_vs_f32_bluestein_port": dstat(ref32, ref32b),
            "reference_cufft_vs_opencv_published": {"300x194": {"maxdiff": 1, "diff_sum": 44}, "2400x1552": {"maxdiff": 1, "diff_sum": 17631}}.get(f"{pw}x{ph}"),
            "note": "all ports use the reference's float32 eigenvalue tables; f64 / f32 = precision of the two 1-D transforms (oracle/sc_oracle.c "
                    "sco_solve_dst3).  Two float32 transforms differ from each other as much as either differs from the double one, and "
                    "as much as the reference's float32 cuFFT path differed from OpenCV's float32 dft (published)"}
        print(json.dumps(row), flush=True)
    inst.destroy()


def new_size_leg(capi, seed=4):
    """What a ROI size the instance has not seen costs.  The reference rebuilds its per-size state inside EVERY run()
    (SeamlessClone::init_resize, seamlessClone_imp.cpp:1073-1116: DST matrix / eigenvalue tables on the device, cuBLAS pointer
    arrays), so its published times include it; here per-size state (multigrid hierarchy + bottom solver matrices, chirp /
    transform tables of the direct solve, float-table correction tables) is cached, and a caller whose mask changes every
    frame meets a new size on most calls.  Device-resident images, synchronous calls, host wall time per call:
      * fresh: the very first call of a fresh instance at the reference's four published patch sizes (arena allocation,
        per-size state, code-object load of the kernels it touches: everything);
      * stream: 64 clones with pseudo-random ROI sizes drawn from [100, 720]^2 (the direct solve's range) and from [1000, 2400]^2 (multigrid) on one instance whose
        arena has been grown by one call at the largest size -- every call is a size the instance has never seen; right
        after each, the same call again (the size is cached now: steady state).  median / p95 of both and of the ratio."""
    import numpy as np
    rng = np.random.default_rng(seed)
    big = 2402
    dstw = big + 64
    noise_p = rng.integers(0, 256, (big, big, 3), dtype=np.uint8)
    noise_d = np.clip(128.0 + rng.normal(0.0, 14.0, (dstw, dstw, 3)), 0, 255).astype(np.uint8)
    mask = np.full((big, big), 255, np.uint8)
    out = {}

    def call(inst, dev, pw, ph):
        d_face, d_body, d_keep, d_mask = dev
        # a pw x ph patch / mask and a (pw + 64) x (ph + 64) destination as views of the big device images (row steps of the big ones)
        inst.copy_d2d_async(d_body, d_keep, noise_d.nbytes)
        inst.sync()
        t0 = time.perf_counter()
        rc = inst.L.sc_hip_run_device(inst.h, d_face, pw, ph, 3 * big, d_body, pw + 64, ph + 64, 3 * dstw, d_mask, pw, ph, big,
                                      (pw + 64) // 2, (ph + 64) // 2, True)
        dt = (time.perf_counter() - t0) * 1e3
        if rc not in (capi.SC_OK, capi.SC_ERR_NOT_CONVERGED):
            raise capi.SeamlessCloneError(rc, f"new_size leg, {pw}x{ph}")
        return dt

    def upload(inst):
        return inst.to_device(noise_p), inst.to_device(noise_d), inst.to_device(noise_d), inst.to_device(mask)

    fresh = {}
    for pw, ph in ((154, 100), (300, 194), (592, 592), (2400, 1552)):
        inst = capi.Instance(0)
        try:
            dev = upload(inst)
            first = call(inst, dev, pw, ph)
            again = sorted(call(inst, dev, pw, ph) for _ in range(5))[2]
            fresh[f"{pw}x{ph}"] = {"first_call_ms": round(first, 3), "steady_ms": round(again, 3), "method": int(inst.info().method)}
            for d in dev:
                inst.free(d)
        finally:
            inst.destroy()
    # ... and a 4096 x 4096 ROI (its own images: 50 MB each)
    big4 = 4098
    p4 = rng.integers(0, 256, (big4, big4, 3), dtype=np.uint8)
    d4 = np.clip(128.0 + rng.normal(0.0, 14.0, (big4 + 64, big4 + 64, 3)), 0, 255).astype(np.uint8)
    m4 = np.full((big4, big4), 255, np.uint8)
    inst = capi.Instance(0)
    try:
        dev4 = (inst.to_device(p4), inst.to_device(d4), inst.to_device(d4), inst.to_device(m4))

        def call4():
            inst.copy_d2d_async(dev4[1], dev4[2], d4.nbytes)
            inst.sync()
            t0 = time.perf_counter()
            rc = inst.L.sc_hip_run_device(inst.h, dev4[0], big4, big4, 3 * big4, dev4[1], big4 + 64, big4 + 64, 3 * (big4 + 64), dev4[3], big4, big4, big4,
                                          (big4 + 64) // 2, (big4 + 64) // 2, True)
            dt = (time.perf_counter() - t0) * 1e3
            if rc not in (capi.SC_OK, capi.SC_ERR_NOT_CONVERGED):
                raise capi.SeamlessCloneError(rc, "new_size leg, 4096^2")
            return dt
        first = call4()
        again = sorted(call4() for _ in range(5))[2]
        fresh["4098x4098"] = {"first_call_ms": round(first, 3), "steady_ms": round(again, 3), "method": int(inst.info().method)}
        for d in dev4:
            inst.free(d)
    finally:
        inst.destroy()
    del p4, d4, m4
    out["fresh_instance"] = fresh

    inst = capi.Instance(0)
    try:
        dev = upload(inst)
        call(inst, dev, 2402, 2402)                 # grow the arena once: from here on a new size costs its per-size state only
        call(inst, dev, 902, 902)
        call(inst, dev, capi.SC_AUTO_DIRECT_MAX + 2, capi.SC_AUTO_DIRECT_MAX + 2)      # ... and the direct solve's work planes (the two calls above ran the cycles)
        for name, lo, hi in (("roi_100_720", 100, capi.SC_AUTO_DIRECT_MAX), ("roi_1000_2400", 1000, 2400)):
            first, steady, seen = [], [], set()
            while len(first) < 64:
                pw, ph = int(rng.integers(lo, hi + 1)) + 2, int(rng.integers(lo, hi + 1)) + 2
                if (pw, ph) in seen:
                    continue
                seen.add((pw, ph))
                first.append(call(inst, dev, pw, ph))
                assert inst.info().new_size == 1
                steady.append(min(call(inst, dev, pw, ph), call(inst, dev, pw, ph)))
                assert inst.info().new_size == 0
            ratio = sorted(f / s for f, s in zip(first, steady))
            fs, ss = sorted(first), sorted(steady)
            out[name] = {"clones": len(first), "first_call_ms": {"median": round(fs[32], 3), "p95": round(fs[60], 3), "max": round(fs[-1], 3)},
                         "steady_ms": {"median": round(ss[32], 3), "p95": round(ss[60], 3)},
                         "first_over_steady": {"median": round(ratio[32], 3), "p95": round(ratio[60], 3), "max": round(ratio[-1], 3)}}
        for d in dev:
            inst.free(d)
    finally:
        inst.destroy()
    out["note"] = ("device-resident images, synchronous sc_hip_run_device, host wall time per call; first = a ROI size the instance has "
                   "never seen, steady = the same call repeated (best of 2); the reference rebuilds its per-size state in every call "
                   "(seamlessClone_imp.cpp:1073-1116), here it is built on the device once per size and cached")
    return out


def mixed_sizes_leg(capi, lo=1000, hi=1100, n=64, streams=2, group=0, reps=15, seed=2025, sizes=None):
    """A batch whose members all have DIFFERENT ROI sizes -- what real clones produce: a mask box per face, per frame -- through the
    native pool with its default grouping (size classes share one set of launches: csrc/sc_ragged.cpp), beside (a) the same list one
    clone at a time on 8 streams (what rounds 1-4 did with such a batch) and (b) n same-size clones of the list's mean size (the
    ceiling).  Device-resident images, destinations refreshed inside the step, wall time of pool.run (median of `reps`)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = [(int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))) for _ in range(n)]
    else:
        n = len(sizes); lo = min(min(s_) for s_ in sizes); hi = max(max(s_) for s_ in sizes)
    Hd = Wd = hi + 64
    yy, xx = np.mgrid[0:Hd, 0:Wd].astype(np.float32)
    dst = np.clip((128.0 + 60.0 * np.sin(2 * np.pi * xx / Wd) * np.cos(2 * np.pi * yy / Hd))[:, :, None] +
                  12.0 * rng.standard_normal((Hd, Wd, 3), dtype=np.float32), 0, 255).astype(np.uint8)
    xx = np.arange(hi + 2, dtype=np.float32)[None, :, None]
    patch = np.clip(110.0 + 50.0 * np.cos(3 * np.pi * xx / hi) + 20.0 * rng.standard_normal((hi + 2, hi + 2, 3), dtype=np.float32), 0, 255).astype(np.uint8)
    g, k = capi.plan_groups_pool(sizes, group, streams)      # (group 0 = SC_POOL_GROUP_AUTO: the pool sizes its groups itself, seamlessclone_hip.h)
    planned = sorted((g.count(q) for q in set(g)), reverse=True)
    ci = k.index(2) if 2 in k else 0           # a member that keeps its own hierarchy inside its class (kind 2): the byte-for-byte check

    def run(streams_, group_, sz, check=False, unseen=False):
        pool = capi.Pool(0, streams=streams_, group=group_)
        try:
            inst = pool.instances[0]
            jobs = pool.make_jobs(len(sz)); keep = []
            d0 = inst.to_device(dst)
            for j, (W, H) in zip(jobs, sz):
                f, b, m = inst.to_device(np.ascontiguousarray(patch[:H + 2, :W + 2])), inst.to_device(dst), inst.to_device(np.full((H + 2, W + 2), 255, np.uint8))
                keep += [f, b, m]
                j.face, j.face_cols, j.face_rows, j.face_step = f, W + 2, H + 2, 3 * (W + 2)
                j.body, j.body_cols, j.body_rows, j.body_step = b, Wd, Hd, 3 * Wd
                j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, W + 2, H + 2, W + 2
                j.centerX, j.centerY, j.body_restore = Wd // 2, Hd // 2, d0
            for _ in range(4):
                pool.run(jobs, device_resident=True)
            ts = []
            for _ in range(reps):
                if unseen:
                    capi.plan_cache_clear()      # every member pays its plan again, as in a stream of sizes never seen before
                t0 = time.perf_counter()
                pool.run(jobs, device_resident=True)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            first = inst.from_device(keep[3 * ci + 1], dst.shape) if check else None
            for p_ in keep + [d0]:
                inst.free(p_)
        finally:
            pool.close()
        t = ts[len(ts) // 2]
        return {"ms_per_step": round(t * 1e3, 3), "Mpix_per_s": round(sum(w * h for w, h in sz) / t / 1e6, 1)}, first
    mixed, a = run(streams, group, sizes, check=True)
    ones, b = run(8, 1, sizes, check=True)
    fresh, _ = run(streams, group, sizes, unseen=True)
    mean = int(round(float(np.sqrt(np.mean([w * h for w, h in sizes])))))
    same, _ = run(streams, group, [(mean, mean)] * n)
    of16, _ = run(streams, 16, sizes) if group != 16 else (mixed, None)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"roi_range": [lo, hi], "clones": n, "streams": streams, "group": group if group else "auto (SC_POOL_GROUP_AUTO)", "planned_groups": planned,
            "with_groups_of_16": of16,
            "Mpix_per_s": mixed["Mpix_per_s"], "ms_per_step": mixed["ms_per_step"],
            "planner_memo_cleared_every_step": fresh, "one_clone_at_a_time_8_streams": ones, "same_size_%d" % mean: same,
            "ratio_to_same_size": round(mixed["Mpix_per_s"] / same["Mpix_per_s"], 3),
            "members_on_a_deeper_hierarchy_than_solo": int(sum(1 for x in k if x == 3)),
            "first_member_vs_its_solo_clone": {"member": ci, "maxdiff": int(d.max()), "percent_differing": round(float((d > 0).mean() * 100), 4)},
            "note": "64 clones with 64 different ROI sizes, random in [%d, %d]^2, through the pool with automatic group sizes (the same-size ceiling too): members of one size class (same "
                    "hierarchy depth and bottom solve, widths and heights within 2x; the pool hands its jobs to the planner largest first) share one set of solver launches through a per-member geometry "
                    "table (nothing of it is kept between calls; only the size plans are memoised per size -- planner_memo_cleared_every_step forgets those too); every member's bytes are its solo run's whenever the group takes the solo run's cycle count -- except the leftovers of a class "
                    "one level shallower, which ride along on the deeper hierarchy (counted above; within one grey level of their solo runs)" % (lo, hi)}


def c5_projection_leg(capi, reps=8):
    """BASELINE config 5 at N = 8, projected from ONE GPU (no 8-GPU node is available to this build): at N = 8 every rank clones 8 of
    the 64 independent 1024 x 1024 images with no interaction whatsoever, so a rank's time is this GPU's time for ITS 8 images and the
    job's throughput is 8 x that shard's (max over ranks = any rank: the shards are alike).  Times rank 0's shard -- images 0, 8, ...,
    56 of --config c5 -- under every stream x group split of 8 images, and all 64 images on this GPU (N = 1: 2 streams x groups of
    32) for the strong-scaling efficiency to expect."""
    import numpy as np
    gen = BatchSynth(1024, 3000)

    def run(ids, streams, group):
        pool = capi.Pool(0, streams=streams, group=group)
        try:
            inst = pool.instances[0]
            jobs = pool.make_jobs(len(ids)); keep = []
            for j, k in zip(jobs, ids):
                dst, patch, mask, cx, cy = gen.image(k)
                f, b0, b, m = inst.to_device(patch), inst.to_device(dst), inst.to_device(dst), inst.to_device(mask)
                keep += [f, b0, b, m]
                j.face, j.face_cols, j.face_rows, j.face_step = f, patch.shape[1], patch.shape[0], 3 * patch.shape[1]
                j.body, j.body_cols, j.body_rows, j.body_step = b, dst.shape[1], dst.shape[0], 3 * dst.shape[1]
                j.mask, j.mask_cols, j.mask_rows, j.mask_step = m, mask.shape[1], mask.shape[0], mask.shape[1]
                j.centerX, j.centerY, j.body_restore = cx, cy, b0
            for _ in range(4):
                pool.run(jobs, device_resident=True)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                pool.run(jobs, device_resident=True)
                ts.append(time.perf_counter() - t0)
            for p_ in keep:
                inst.free(p_)
        finally:
            pool.close()
        ts.sort()
        return ts[len(ts) // 2]
    shard = list(range(0, 64, 8))
    splits = {}
    for streams, group in ((1, 8), (2, 4), (4, 2), (8, 1)):
        t = run(shard, streams, group)
        splits["%dx%d" % (streams, group)] = {"ms": round(t * 1e3, 3), "Mpix_per_s": round(8 * 1024 * 1024 / t / 1e6, 1)}
    best = max(splits, key=lambda k: splits[k]["Mpix_per_s"])
    t64 = run(list(range(64)), 2, 32)
    n1 = 64 * 1024 * 1024 / t64 / 1e6
    proj = 8 * splits[best]["Mpix_per_s"]
    return {"Mpix_per_s": round(proj, 1), "shard_of_8": {"images": shard, "split_streams_x_group": best, "Mpix_per_s": splits[best]["Mpix_per_s"], "all_splits": splits},
            "n1_all_64_images": {"split_streams_x_group": "2x32", "ms": round(t64 * 1e3, 3), "Mpix_per_s": round(n1, 1)},
            "strong_scaling_efficiency_expected_at_8": round(proj / (8 * n1), 3),
            "note": "config 5 at N = 8 = 8 images per GPU, no communication: 8 x (this GPU's throughput on rank 0's shard, best stream x group "
                    "split); the efficiency below 1 is a per-GPU batch-size effect (8 images fill the chip less well than 64), not a "
                    "communication one -- NOT a measured 8-GPU number"}


def launch_ranks(args):
    """--gpus N > 1 without a launcher: one fresh child process per rank, started BEFORE this process touches HIP
    (it never does: the library is only loaded by the children).  Rank 0's JSON line is forwarded."""
    from seamlesscloneoptimization_amd.batch import spawn_ranks
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    rc, out0 = spawn_ranks(args.gpus, cmd)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if args.reference_table:
        return reference_table(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # The HIP library must be loaded before anything pulls in torch's bundled ROCm runtime.
    from seamlesscloneoptimization_amd import capi
    capi.load()   # bind /opt/rocm's HIP runtime now; torch (gloo only, N>1) is imported later inside Comm()
    from seamlesscloneoptimization_amd.batch import Comm, timed_region
    methods = {"mg": capi.SC_METHOD_MULTIGRID, "sor": capi.SC_METHOD_SOR, "rbgs": capi.SC_METHOD_RBGS,
               "jacobi": capi.SC_METHOD_JACOBI, "auto": capi.SC_METHOD_AUTO, "fft": capi.SC_METHOD_FFT, "dst": capi.SC_METHOD_DST}

    comm = Comm()
    if comm.world != args.gpus:
        args.gpus = comm.world
    # CPU placement before the library starts any thread (pool workers, copy helpers inherit the mask)
    ndev0 = max(1, capi.device_count())
    cores = pin_rank_to_its_cores(comm.local_rank, comm.world, args.affinity, capi.device_pci_bus_id(comm.local_rank % ndev0))
    import numpy as np
    from seamlesscloneoptimization_amd.batch import shard_indices
    image_ids = None
    if args.config == "c5":
        # BASELINE config 5 as written: 64 independent 1024 x 1024 clones, image i on rank i mod N (8 per GPU at N = 8), no
        # collective; one step = every rank clones its shard once; value = 64 x 1024^2 x steps / max-over-ranks time (STRONG scaling)
        args.roi = 1024
        image_ids = shard_indices(64, comm.rank, comm.world)
        args.batch = len(image_ids)
        if args.batch == 0:
            sys.exit("bench.py --config c5: more ranks than images")
        args.streams = max(1, min(args.streams, args.batch))
        args.group = max(1, min(args.group if args.group else 32, -(-args.batch // args.streams)))
    if not args.group:
        args.group = 16

    ndev = capi.device_count()
    if ndev < 1:
        sys.exit("bench.py: no MI355X visible (there is no CPU fallback)")
    opts = dict(method=methods[args.method], flags=(capi.SC_FLAG_EXACT_TABLES if args.exact_tables else 0) | args.extra_flags)
    if args.method == "sor":
        opts.update(tol=2e-5, max_sweeps=200000, check_every=64)
    if args.sweeps_per_launch:
        opts.update(sweeps_per_launch=args.sweeps_per_launch)
    streams = max(1, min(args.streams, args.batch))
    group = max(1, min(args.group, args.batch))
    streams = max(1, min(streams, (args.batch + group - 1) // group))
    # % ndev only matters when rehearsing N ranks on a 1-GPU box
    pool = capi.Pool(comm.local_rank % ndev, streams, group=group, **opts)      # native C++ workers, one per HIP stream
    inst = pool.instances[0]

    W = H = args.roi
    jobs = []
    cjobs = pool.make_jobs(args.batch)          # one C job per image: D2D restore + device-resident clone
    gen = BatchSynth(args.roi, 3000 + comm.rank if image_ids is not None else 1001 + 7919 * comm.rank)
    for b in range(args.batch):
        if b == 0 and comm.rank == 0 and image_ids is None:
            dst, patch, mask, cx, cy = synth(args.roi, 0)      # SURVEY 8d's image itself: the CPU baseline and the parity figures use it
        else:
            dst, patch, mask, cx, cy = gen.image(image_ids[b] if image_ids is not None else b)
        j = dict(host=(dst, patch, mask, cx, cy) if b in (0, args.batch - 1) else None, f=inst.to_device(patch), fs=patch.shape[:2],
                 b0=inst.to_device(dst), b=inst.to_device(dst), n=dst.nbytes, bs=dst.shape[:2],
                 m=inst.to_device(mask), ms=mask.shape[:2], cx=cx, cy=cy)
        jobs.append(j)
        c = cjobs[b]
        c.face, c.face_cols, c.face_rows, c.face_step = j["f"], patch.shape[1], patch.shape[0], 3 * patch.shape[1]
        c.body, c.body_cols, c.body_rows, c.body_step = j["b"], dst.shape[1], dst.shape[0], 3 * dst.shape[1]
        c.mask, c.mask_cols, c.mask_rows, c.mask_step = j["m"], mask.shape[1], mask.shape[0], mask.shape[1]
        c.centerX, c.centerY, c.body_restore = cx, cy, j["b0"]

    def clone(i, j, sync=False):
        i.copy_d2d_async(j["b"], j["b0"], j["n"])
        i.run_device(j["f"], j["fs"], j["b"], j["bs"], j["m"], j["ms"], j["cx"], j["cy"], sync=sync)

    def step():
        pool.run(cjobs, device_resident=True)   # returns when the whole batch is complete

    def sync_all():
        for i in pool.instances:
            i.sync()

    step()                                   # set-up, not a warm-up step: sizes every instance's device arena for this ROI (hipMalloc)
    for _ in range(args.warmup):
        step()
    t_rank = []
    elapsed = timed_region(comm, sync_all, lambda: [step() for _ in range(args.steps)], t_rank)
    per_rank_ms = [round(t * 1e3, 3) for t in comm.gather(t_rank[0])]          # this rank's own elapsed time, every rank's on rank 0
    group_cycles = max(i.info().sweeps for i in pool.instances)
    devices = comm.gather(float(inst.info().device))
    # the same timed step with float32 storage throughout (field between the level-0 launches, right-hand side, level 1): what the
    # 16-bit / float16 storage formats of the default are worth, timed by the same clock
    value_f32 = None
    if args.method == "mg" and not args.no_float32_leg:
        pool.set_solver(flags=opts["flags"] | capi.SC_FLAG_FLOAT_FIELD | capi.SC_FLAG_FLOAT_RHS)
        for _ in range(max(1, min(args.warmup, 2))):
            step()
        el32 = timed_region(comm, sync_all, lambda: [step() for _ in range(args.steps)])
        value_f32 = (el32, max(i.info().sweeps for i in pool.instances))
        pool.set_solver(flags=opts["flags"])
        step()                                # back on the default's fields (level 1 re-zeroed) before anything else is measured
    # ... and the same step without its destination restores: S steps, each cloning into its OWN pre-resident copies of the 32
    # destinations (the clone is in place; `value` refreshes the same 32 images from pristine copies inside every step -- 0.5 GB of
    # device copies, 2-3 % of the step -- so that any number of steps runs in fixed memory).  Reported beside `value`, never as it.
    value_fresh = None
    if args.method == "mg" and image_ids is None and not args.no_fresh_leg:
        S = max(1, min(args.steps, 16))
        fresh_sets, fresh_jobs = [], []
        for k in range(S):
            cj = pool.make_jobs(args.batch)
            bodies = []
            for b, j in enumerate(jobs):
                nb = inst.malloc(j["n"])
                inst.copy_d2d_async(nb, j["b0"], j["n"])
                bodies.append(nb)
                c, src = cj[b], cjobs[b]
                c.face, c.face_cols, c.face_rows, c.face_step = src.face, src.face_cols, src.face_rows, src.face_step
                c.body, c.body_cols, c.body_rows, c.body_step = nb, src.body_cols, src.body_rows, src.body_step
                c.mask, c.mask_cols, c.mask_rows, c.mask_step = src.mask, src.mask_cols, src.mask_rows, src.mask_step
                c.centerX, c.centerY, c.body_restore = src.centerX, src.centerY, None
            fresh_sets.append(bodies); fresh_jobs.append(cj)
        inst.sync()
        step()                                # warm (restoring form)
        elf = timed_region(comm, sync_all, lambda: [pool.run(cj, device_resident=True) for cj in fresh_jobs])
        same = np.array_equal(inst.from_device(fresh_sets[-1][0], jobs[0]["host"][0].shape), inst.from_device(jobs[0]["b"], jobs[0]["host"][0].shape))
        value_fresh = (elf, S, bool(same))
        for bodies in fresh_sets:
            for nb in bodies:
                inst.free(nb)
    dst, patch, mask, cx, cy = jobs[0]["host"]
    out = inst.from_device(jobs[0]["b"], dst.shape)
    if not all(i.info().converged for i in pool.instances) or np.array_equal(out, dst):
        sys.exit("bench.py: the clone did not converge / did not modify the destination")

    # ---- roofline of the dominant kernel AS TIMED: the level-0 multigrid launch of a group (3 x group channels), on the
    #      fields the last group of the timed region left on instance 0; HIP events on the library's stream around
    #      back-to-back launches.  (The isolated launches run the same kernel under a second symbol -- template tag -- so
    #      the rocprofv3 statistics of this command keep them apart from the launches inside the clones.)
    unknowns = (W - 2) * (H - 2) * 3
    grp_ch = inst.field_shape()[0] if args.method == "mg" else 3
    ms_c0 = inst.time_cycle0(args.kernel_launches) if args.method == "mg" else None
    # ... and the other three level-0 launches a solve is made of, each under its own tagged symbol: the launch-weighted figure
    # of what the timed step runs on level 0 (first launch, full cycle, full cycle before the judged one, judged cycle)
    ms_forms = None
    if args.method == "mg" and not (opts["flags"] & (capi.SC_FLAG_FLOAT_FIELD | capi.SC_FLAG_FLOAT_L1 | capi.SC_FLAG_FLOAT_RHS)):
        nl = max(10, args.kernel_launches // 4)
        ms_forms = {"first": inst.time_cycle0_form(3, nl), "full": ms_c0, "full_before_judged": inst.time_cycle0_form(1, nl),
                    "judged_output": inst.time_cycle0_form(2, nl)}

    # the last image of the batch once more, alone: what the pool (groups, several streams) wrote must be the clone's result
    jl = jobs[-1]
    pooled_last = inst.from_device(jl["b"], jl["host"][0].shape)
    clone(inst, jl, sync=True)
    alone_last = inst.from_device(jl["b"], jl["host"][0].shape)
    if int(np.abs(pooled_last.astype(np.int16) - alone_last.astype(np.int16)).max()) > 1:
        sys.exit("bench.py: a pooled clone differs from the same clone run alone by more than one grey level")
    # one synchronous clone alone on the GPU for the per-stage hipEvent breakdown (the first call re-sizes the instance
    # from a group's 3n channels to 3; the second is the steady state)
    clone(inst, jobs[0], sync=True)
    clone(inst, jobs[0], sync=True)
    info = inst.info()
    # ... and the same clone without the stage marks (each is an event in the stream with a ~5 us bubble behind it): the
    # un-instrumented device time of ONE 2048^2 clone -- BASELINE config 3 as written -- by hipEvents, and the wall time of the call
    inst.set_solver(flags=opts["flags"] | capi.SC_FLAG_NO_STAGE_MARKS)
    solo_dev, solo_wall = [], []
    for _ in range(12):
        inst.copy_d2d_async(jobs[0]["b"], jobs[0]["b0"], jobs[0]["n"])
        inst.sync()
        t0 = time.perf_counter()
        inst.run_device(jobs[0]["f"], jobs[0]["fs"], jobs[0]["b"], jobs[0]["bs"], jobs[0]["m"], jobs[0]["ms"], jobs[0]["cx"], jobs[0]["cy"], sync=True)
        solo_wall.append((time.perf_counter() - t0) * 1e3)
        solo_dev.append(inst.info().ms_device_total)
    inst.set_solver(flags=opts["flags"])
    # the launch-bound part of a cycle (levels 2 .. bottom .. 2: five dependent launches per cycle for this ROI, seven before k_mg_tail) as plain launches
    # and as replays of one captured HIP graph, on the hierarchy this clone left
    chain = None
    if args.method == "mg":
        try:
            ce, cg, cn = inst.time_coarse_chain(50)
            chain = {"dependent_launches": cn, "us_per_pass_plain_launches": round(ce * 1e3, 2), "us_per_pass_hip_graph_replay": round(cg * 1e3, 2),
                     "note": "levels 2 .. bottom .. 2 of one cycle, 50 passes back to back, hipEvents on the instance's stream; a graph replay is "
                             "launched per pass (hipGraphLaunch), the plain form enqueues the same kernels one by one"}
        except capi.SeamlessCloneError as e:
            chain = {"error": str(e)}
        if isinstance(chain, dict) and "error" not in chain:
            try:        # the level above the bottom + the bottom in one launch: shader-clock cycles between its phase boundaries (channel 0)
                ph = sorted(inst.time_tail_phases() for _ in range(5))[2]
                chain["k_mg_tail_phase_cycles"] = dict(zip(["entry_to_rhs_in_registers", "pre_smoothing", "residual_and_restriction", "product_1", "product_2",
                                                            "product_3", "product_4", "interpolation", "post_smoothing", "stores"], ph))
            except capi.SeamlessCloneError:
                pass        # this hierarchy runs the three launches (level above the bottom too large for the registers)
    solo_dev.sort(); solo_wall.sort()
    single_clone = {"ms": round(solo_dev[len(solo_dev) // 2], 4), "Mpix_per_s": round(W * H / (solo_dev[len(solo_dev) // 2] * 1e-3) / 1e6, 1),
                    "ms_min": round(solo_dev[0], 4), "wall_ms_median": round(solo_wall[len(solo_wall) // 2], 4), "cycles": int(inst.info().sweeps),
                    "coarse_chain": chain,
                    "note": "ONE clone alone on the GPU (BASELINE config 3 as written), images resident in HBM: median of 12 of the hipEvent "
                            "time from the first to the last kernel of the clone, no stage marks in between (SC_FLAG_NO_STAGE_MARKS); "
                            "wall = host time of the synchronous call"}
    # the same clone with the exact tables (parity pair of the CPU baseline leg)
    if args.method == "mg":
        inst.set_solver(flags=opts["flags"] ^ capi.SC_FLAG_EXACT_TABLES)
        clone(inst, jobs[0], sync=True)
        other_tables = inst.from_device(jobs[0]["b"], dst.shape)
        inst.set_solver(flags=opts["flags"])
    else:
        other_tables = out
    out_float, out_exact = (other_tables, out) if args.exact_tables else (out, other_tables)

    # ---- the reference's direct solve on the matrix cores (SC_METHOD_DST, sc_dst.hip): four double-precision products with the
    #      DST matrix per channel, 2 n^3 multiply-adds each; timed as the solve stage of one synchronous clone
    inst.set_solver(method=capi.SC_METHOD_DST)
    clone(inst, jobs[0], sync=True)                  # builds the DST matrices for this size
    clone(inst, jobs[0], sync=True)
    di = inst.info()
    out_dst = inst.from_device(jobs[0]["b"], dst.shape)
    # ---- the reference's DEFAULT back-end: FFT-based direct solve in float32 (SC_METHOD_FFT, sc_fft.hip)
    inst.set_solver(method=capi.SC_METHOD_FFT)
    clone(inst, jobs[0], sync=True)                  # builds the chirp / transform tables for this size
    clone(inst, jobs[0], sync=True)
    fi = inst.info()
    out_fft = inst.from_device(jobs[0]["b"], dst.shape)
    inst.set_solver(method=methods[args.method])
    wp = hp = 2 * (-(-((W - 2 + 1) // 2) // 128) * 128)       # two parity halves, each padded to 128
    dst_flop = 3 * (2.0 * hp * wp * wp + 2.0 * hp * hp * wp)      # the even/odd fold halves the plain matrix form's 2 n^3 per product
    dd = np.abs(out_dst.astype(np.int16) - out_float.astype(np.int16))
    roofline_dst = {"bound": "mfma", "kernel": "k_dgemm<EPI> x 4 on parity-folded halves (v_mfma_f64_16x16x4_f64, 128x128x16 LDS tiles, 8 waves) + fold / unfold, SC_METHOD_DST",
                    "achieved": round(dst_flop / (di.ms_solve * 1e-3) / 1e12, 2), "peak": 78.6, "unit": "TFLOP/s",
                    "frac": round(dst_flop / (di.ms_solve * 1e-3) / 1e12 / 78.6, 4), "traffic": None,
                    "ms_solve": round(di.ms_solve, 4), "ms_device_total": round(di.ms_device_total, 4),
                    "flop_per_clone": int(dst_flop),
                    "vs_default_path": {"maxdiff": int(dd.max()), "percent_differing": round(float((dd > 0).mean() * 100), 4)},
                    "note": "the reference's own algorithm (direct DST, float tables); the DST matrix's mirror symmetry splits every product "
                            "into two of half the size (flop_per_clone counts what is executed: half of the plain matrix form); peak = MI355X "
                            "FP64 matrix spec; solve stage of one clone by hipEvents (includes the fold / unfold kernels)"}

    nfft = 2                                   # the library's choice (sc_fft.hip fft_len): the shortest r 2^k >= 2n - 1, r in {1, 3, 5}
    while nfft < 2 * (W - 2) - 1:
        nfft *= 2
    for r_ in (3, 5):
        k_ = 4
        while (r_ << k_) < nfft:
            if (r_ << k_) >= 2 * (W - 2) - 1:
                nfft = r_ << k_
                break
            k_ += 1
    import math
    lg = math.log2(nfft)
    fft_flop = 3.0 * 4 * (W - 2) * 2 * 5.0 * nfft * lg            # 4 DST passes x rows x 2 FFTs x 5 M log2 M, 3 channels
    fft_bytes = 3.0 * (W - 2) * (H - 2) * 4 * (2 + 2 + 2 + 2 + 2 + 1)   # five launches read + write a float per unknown; the first reads F, the last writes U
    df = np.abs(out_fft.astype(np.int16) - out_float.astype(np.int16))
    roofline_fft = {"bound": "hbm", "kernel": "k_fft_dst<0|1|2> (chirp-z DST-I over a power-of-two FFT in LDS, one workgroup per row) + 2 x k_fft_transpose, SC_METHOD_FFT",
                    "achieved": round(fft_bytes / (fi.ms_solve * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(fft_bytes / (fi.ms_solve * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                    "ms_solve": round(fi.ms_solve, 4), "ms_device_total": round(fi.ms_device_total, 4), "fft_length": nfft,
                    "flop_per_clone": int(fft_flop), "TFLOP_per_s_fp32": round(fft_flop / (fi.ms_solve * 1e-3) / 1e12, 2),
                    "algorithmic_bytes_per_clone": int(fft_bytes),
                    "vs_default_path": {"maxdiff": int(df.max()), "percent_differing": round(float((df > 0).mean() * 100), 4)},
                    "note": "the reference's default algorithm (direct DST via FFTs, float32, float tables); solve stage of one clone by "
                            "hipEvents; the launches are bound by LDS bandwidth and float32 butterflies, not by HBM (frac = algorithmic "
                            "bytes of the five launches / time / 8 TB/s)"}

    # ---- the drop-in call itself: pageable host images in, result in the caller's image (PCIe inclusive).  --host-calls timed
    #      calls after two untimed ones (the first sizes the pinned staging: hipHostMalloc of ~40 MB), the destination restored on
    #      the host between calls; median / p95 / min of the call and of its stages
    body = dst.copy()
    first_ms = []
    for _ in range(2):
        body[...] = dst
        t0 = time.perf_counter()
        inst.run(patch, body, mask, cx, cy)
        first_ms.append(round((time.perf_counter() - t0) * 1e3, 4))
    host_result = body.copy()
    flags0 = inst.get_solver().flags

    def host_calls(n, flags):
        inst.set_solver(flags=flags)
        out = []
        for _ in range(n):
            body[...] = dst
            t0 = time.perf_counter()
            inst.run(patch, body, mask, cx, cy)
            t_host = (time.perf_counter() - t0) * 1e3
            hi = inst.info()
            out.append((t_host, hi.ms_h2d, hi.ms_device_total, hi.ms_d2h, hi.ms_call))
        if not np.array_equal(body, host_result):
            raise SystemExit("bench: the drop-in call's result changed between calls")
        return out
    # the statistics of the call: no stage marks inside it (each is an event in the stream with a ~5 us bubble behind it); its stages: a
    # second, shorter series with the marks
    calls = host_calls(max(3, args.host_calls), flags0 | capi.SC_FLAG_NO_STAGE_MARKS)
    marked = host_calls(max(3, args.host_calls // 3), flags0)
    rows_ret = host_calls(max(3, args.host_calls // 2), flags0 | capi.SC_FLAG_NO_STAGE_MARKS | capi.SC_FLAG_ROWS_RETURN)
    inst.set_solver(flags=flags0)

    def stat(k, series=None):
        v = sorted(c[k] for c in (series if series is not None else calls))
        return {"median": round(v[len(v) // 2], 4), "p95": round(v[min(len(v) - 1, int(round(0.95 * (len(v) - 1))))], 4), "min": round(v[0], 4)}
    st_call = stat(0)
    pcie = {"call_ms": st_call["median"], "call_ms_p95": st_call["p95"], "call_ms_min": st_call["min"], "calls_timed": len(calls),
            "first_two_calls_ms": first_ms, "call_ms_with_stage_marks": stat(0, marked), "call_ms_rows_return": stat(0, rows_ret),
            "h2d_ms": stat(1, marked), "device_ms": stat(2, marked), "d2h_ms": stat(3, marked), "stream_ms": stat(4),
            "Mpix_per_s_inclusive": round(W * H / (st_call["median"] * 1e-3) / 1e6, 1),
            "note": "my_seamlessclone_api_imp_run on pageable numpy images, one clone, one instance: mask, patch rows and destination rows "
                    "cross PCIe as linear copies at the caller's row step (no packing: the ROI covers most of every row here), clone, the "
                    "compact ROI back through pinned staging and spliced into the caller's rows -- only ROI bytes are ever written (the "
                    "default since round 5; call_ms_rows_return = the opt-in SC_FLAG_ROWS_RETURN, round 4's default: output bytes written "
                    "into the destination rows on the device, those rows back as one linear copy into the caller's image) (never `value`); call_ms = host wall time of the call (median, SC_FLAG_NO_STAGE_MARKS), stream_ms = hipEvent time "
                    "from the first upload to the last download of the same calls; h2d / device / d2h from a second series with the marks"}

    # ---- sweep kernels named by the north-star, on freshly built float fields of the same images (single clone, 3 channels)
    spl = args.sweeps_per_launch            # 0: library default = fused kernels at their deepest depth
    rb_depth = {0: 4, 1: 0, -1: 1}.get(spl, min(spl, 4))        # sweeps per launch of the red-black kernel
    j_depth = {0: 8, 1: 0, -1: 1, 5: 4, 7: 6}.get(spl, min(spl, 8))
    inst.build_rhs(patch, dst, mask, cx, cy)
    ms_rb = inst.field_time_sweeps(capi.SC_METHOD_RBGS, args.kernel_launches, spl, 1.0)
    ms_j = inst.field_time_sweeps(capi.SC_METHOD_JACOBI, args.kernel_launches, spl, 1.0)
    ms_j1 = inst.field_time_sweeps(capi.SC_METHOD_JACOBI, args.kernel_launches, 1, 1.0)
    src_fp = capi.source_fingerprint()
    rb_bytes = 12.0 * unknowns * (0.5 if rb_depth == 0 else rb_depth)   # plain kernel: one colour per launch
    j_bytes = 12.0 * unknowns * max(j_depth, 1)

    # Counter figures (fabric bytes per launch, bytes per timed step) come from committed rocprofv3 --pmc passes of THIS command
    # (tools/r5_profile.sh -> profiles/r5_pmc_traffic_bench.json).  They are only quoted when the capture matches the run:
    # same ROI / batch / group and the same library sources (capi.source_fingerprint); otherwise `traffic` is null and
    # `traffic_stale` says why -- a stale number is never passed on silently.
    profile, traffic_stale = None, None
    src_now = capi.source_fingerprint()
    for name in ("r5_pmc_traffic_bench.json", "r4_pmc_traffic_bench.json", "r3_pmc_traffic_bench.json"):
        try:
            profile = json.load(open(os.path.join(ROOT, "profiles", name)))
            profile["_file"] = "profiles/" + name
            break
        except Exception:
            pass
    if profile is None:
        traffic_stale = "no committed counter capture under profiles/ (run tools/r5_profile.sh on the GPU box)"
    else:
        why = []
        for key, have in (("roi", args.roi), ("batch", args.batch), ("group", group)):
            if profile.get(key) != have:
                why.append(f"{key} {profile.get(key)} in the capture, {have} in this run")
        if profile.get("source_fingerprint") != src_now:
            why.append(f"library sources changed since the capture (fingerprint {profile.get('source_fingerprint', 'not recorded')} then, {src_now} now)")
        if why:
            traffic_stale = f"{profile['_file']} @ {profile.get('git', '?')} does not describe this run: " + "; ".join(why) + \
                            " -- re-capture with tools/r4_profile.sh"
            profile = None

    def pmc_traffic(symbol, channels):
        """Fabric-side bytes per launch from the committed rocprofv3 PMC passes of this command (2 x FETCH_SIZE + WRITE_SIZE,
        tools/pmc_traffic_bench.py; collected per MI355X_MICROARCH.md's HBM section); None when no capture describes this run."""
        if not profile:
            return None
        for k, v in profile.get("kernels", {}).items():
            if symbol in k and v.get("channels", channels) == channels:
                return v["traffic_bytes_per_launch"]
        return None

    def roof(name, symbol, bytes_per_launch, ms, note, channels=3):
        ach = bytes_per_launch / (ms * 1e-3) / 1e9
        tr = pmc_traffic(symbol, channels)
        r = {"bound": "hbm", "kernel": name, "profiler_symbol": "sc::" + symbol, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "frac_effective": round(ach / HBM_PEAK_GBS, 4),
             "traffic": tr, "frac_traffic": round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr else None,
             "traffic_source": (f"{profile['_file']} @ {profile.get('git', '?')}" if tr else None),
             "traffic_stale": (None if tr else (traffic_stale or "the capture holds no entry for this kernel symbol")),
             "us_per_launch": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": int(bytes_per_launch), "note": note}
        return r

    rb_name = "k_rb_half (one colour, in place)" if rb_depth == 0 else \
        f"k_rb_tb<{rb_depth},8,8> ({rb_depth} fused red-black sweeps per launch, register blocked)"
    j_name = "k_jacobi_roll (register-rolling 5-point, 1 sweep)" if j_depth == 0 else \
        f"k_jacobi_tb<{j_depth},8,8> ({j_depth} fused Jacobi sweeps per launch, register blocked)"

    def cache_note(ch, bytes_per_unknown=12):
        ws = (W - 2) * (H - 2) * ch * bytes_per_unknown
        return "working set %.0f MB %s the 256 MB Infinity Cache" % (ws / 1e6, "fits" if ws < 256e6 else "exceeds")
    rb_sym = "k_rb_half<false, 1>" if rb_depth == 0 else f"k_rb_tb<{rb_depth}, 8, 8, false, false, 8, {1 if rb_depth <= 2 else 2}>"
    j_sym = "k_jacobi_roll<4, 1>" if j_depth == 0 else f"k_jacobi_tb<{j_depth}, 8, 8, 1, {1 if j_depth <= 4 else 2}>"
    roofline_rb = roof(rb_name, rb_sym, rb_bytes, ms_rb,
                       "red-black GS/SOR sweeps alone, single clone (3 channels); algorithmic bytes = 12 B/unknown/channel/sweep x sweeps "
                       "per launch (SURVEY 8d): frac > 1 is 'effective' bandwidth from temporal blocking, frac_traffic is what crosses the fabric; "
                       + cache_note(3))
    # the dominant kernel of the timed region: the fused level-0 multigrid cycle of a GROUP.  Algorithmic bytes per
    # unknown and channel = the SURVEY 8d figures of the operations it fuses: 4 red-black sweeps (4 x 12 B),
    # residual (8 B read) + restricted RHS (1/4 x 4 B written), prolongation (1/4 x 4 B read + 4 B read + 4 B written)
    if args.method == "mg":
        c0_bytes = (4 * 12.0 + 8.0 + 1.0 + 9.0) * (W - 2) * (H - 2) * grp_ch
        roofline = roof(f"k_cycle0<4,8,8,PRO> on a group of {grp_ch // 3} clones = {grp_ch} channels (prolongation + 4 red-black sweeps + "
                        "residual + restriction, one launch)", "k_cycle0<4, 8, 8, true, false, false, %d>" % (19 if (opts["flags"] & capi.SC_FLAG_FLOAT_L1) else 147 if (opts["flags"] & capi.SC_FLAG_FLOAT_FIELD) else 915), c0_bytes, ms_c0,
                        "dominant kernel of the timed region, in the form the timed region launches it; algorithmic bytes = sum of the "
                        "SURVEY 8d figures of the fused operations = 66 B/unknown/channel: frac (= frac_effective) > 1 is 'effective' "
                        "bandwidth from temporal blocking; frac_traffic = counter-measured fabric bytes / time / 8 TB/s.  The isolated "
                        "launches are the tagged 4-sweep composed form WITHOUT the two epilogues some in-step launches carry (cell "
                        "shares of the float-table correction: +7 %; byte output instead of the field store: same duration), on the "
                        "fields the last group left (values are discarded); "
                        + "fields as the launch stores them ("
                        + ("float32 U in / out, " if (opts["flags"] & (capi.SC_FLAG_FLOAT_FIELD | capi.SC_FLAG_FLOAT_L1)) else "16-bit fixed-point U in / out, ")
                        + "float16 right-hand side): "
                        + cache_note(grp_ch, 10 if (opts["flags"] & (capi.SC_FLAG_FLOAT_FIELD | capi.SC_FLAG_FLOAT_L1)) else 6), channels=grp_ch)
        if ms_forms:
            per = (W - 2) * (H - 2) * grp_ch
            # SURVEY 8d per unknown and channel: 12 B per sweep; residual 8 + restriction 1; prolongation 9; output bytes 5 (15 B / pixel)
            alg = {"first": 2 * 12 + 9, "full": 66, "full_before_judged": 66, "judged_output": 2 * 12 + 9 + 5}
            tot_b = sum(alg[k] * per for k in alg)
            tot_ms = sum(ms_forms.values())
            roofline["level0_launches_of_a_solve"] = {
                "us_per_launch": {k: round(v * 1e3, 2) for k, v in ms_forms.items()},
                "profiler_symbols": {"first": "sc::k_cycle0<2, 8, 8, false, false, false, 647>", "full": "sc::k_cycle0<4, 8, 8, true, false, false, 915>",
                                     "full_before_judged": "sc::k_cycle0<4, 8, 8, true, false, false, 467>", "judged_output": "sc::k_cycle0<2, 8, 8, true, false, false, 187>"},
                "algorithmic_bytes_per_unknown_channel": alg,
                "achieved": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "frac_effective": round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "the four level-0 launches of one solve as the timed step runs them (tagged twins of the in-step symbols 646 / 914 / 466 / 186, "
                        "isolated, same fields): launch-weighted algorithmic bytes / time"}
    else:
        roofline = roofline_rb
    roofline_j = roof(j_name, j_sym, j_bytes, ms_j, "the Jacobi stencil named by the north-star, same field, single clone; effective "
                      "bandwidth (temporal blocking); " + cache_note(3))
    roofline_j1 = roof("k_jacobi_roll<4> (5-point sweep, rows rolling through registers, 1 sweep per launch)", "k_jacobi_roll<4, 1>",
                       12.0 * unknowns, ms_j1, "single-sweep Jacobi, single clone: algorithmic == actual traffic; " + cache_note(3))

    # ---- BASELINE config 4: the single-sweep Jacobi kernels at a 4096^2 ROI -- the one configuration whose 604 MB working set
    #      (3 channels x (u, f, u')) exceeds the 256 MB Infinity Cache, i.e. genuinely streams from HBM.  Both forms the library
    #      has: rows rolling through registers (k_jacobi_roll<4>, the default) and the LDS-staged 256 x 32 tile with a 1-pixel halo
    #      that the north-star names (k_jacobi<32>, sc_solver_opts.jacobi_tile_rows = 32).  Counter bytes: profiles/r5_c4_pmc.json.
    roofline_c4 = None
    if not args.no_c4 and args.config == "c3":
        n4 = 4096
        rng4 = np.random.default_rng(44)
        U4 = rng4.integers(0, 256, (3, n4, n4)).astype(np.float32)
        F4 = rng4.integers(-300, 301, (3, n4, n4)).astype(np.float32)
        inst.field_load(U4, F4)
        del U4, F4
        bytes4 = 12.0 * (n4 - 2) * (n4 - 2) * 3
        try:
            c4name = next(n_ for n_ in ("r5_c4_pmc.json", "r4_c4_pmc.json", "r3_c4_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", n_)))
            c4prof = json.load(open(os.path.join(ROOT, "profiles", c4name)))
        except Exception:
            c4prof = None
        c4_stale = None if c4prof and c4prof.get("source_fingerprint") == src_now else \
            ("no profiles/r5_c4_pmc.json" if not c4prof else f"the committed config-4 capture was taken from other library sources "
             f"({c4prof.get('source_fingerprint')} then, {src_now} now): re-capture with tools/r5_profile.sh")
        legs = {}
        for key, rows, sym, label in (("register_rolling", 0, "k_jacobi_roll<4, 1>", "k_jacobi_roll<4>: a wave owns 256 columns x 4 rows, rows y-1..y+4 roll through registers, no LDS, no barrier (default)"),
                                      ("lds_tile_16", 16, "k_jacobi<16, 0>", "k_jacobi<16>: LDS-staged 256 x 16 tile + 1-pixel halo, one barrier"),
                                      ("lds_tile_32", 32, "k_jacobi<32, 0>", "k_jacobi<32>: LDS-staged 256 x 32 tile + 1-pixel halo, one barrier (the form the north-star names)")):
            inst.set_solver(jacobi_tile_rows=rows)
            ms4 = inst.field_time_sweeps(capi.SC_METHOD_JACOBI, max(20, args.kernel_launches), 1, 1.0)
            tr4 = None
            if c4prof and not c4_stale:
                tr4 = next((v["traffic_bytes_per_launch"] for k, v in c4prof.get("kernels", {}).items() if sym in k), None)
            legs[key] = {"kernel": label, "profiler_symbol": "sc::" + sym, "us_per_launch": round(ms4 * 1e3, 2),
                         "achieved": round(bytes4 / (ms4 * 1e-3) / 1e9, 1), "frac": round(bytes4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": tr4, "frac_traffic": round(tr4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr4 else None,
                         "traffic_stale": None if tr4 else c4_stale}
        inst.set_solver(jacobi_tile_rows=0)
        roofline_c4 = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "roi": [n4, n4], "algorithmic_bytes_per_launch": int(bytes4),
                       "working_set_MB": round(bytes4 / 1e6, 1), **legs,
                       "note": "one Jacobi sweep per launch, 3 channels, 12 B per unknown and channel (SURVEY 8d): algorithmic == compulsory "
                               "traffic; HIP events on the library's stream around back-to-back launches; the register-rolling form is the "
                               "default because it is the faster of the two here (no barrier, no LDS round trip for the rows a wave owns)"}

    new_size = None
    if comm.rank == 0 and args.gpus == 1 and not args.no_new_size and args.config == "c3":
        new_size = new_size_leg(capi)
    mixed_sizes = None
    if comm.rank == 0 and args.gpus == 1 and not args.no_mixed_sizes and args.config == "c3":
        mixed_sizes = mixed_sizes_leg(capi)
        wide = mixed_sizes_leg(capi, lo=100, hi=2400, reps=8)      # the hardest list: sizes spread over a factor of 24, four hierarchy depths
        mixed_sizes["wide_range_100_2400"] = {k_: wide[k_] for k_ in wide if k_ not in ("note", "streams", "group", "clones")}
        # the reference's own patch sizes (its published table and size sets: seamlessClone-CUDA README / PDF p3), sixteen clones of each in one batch
        ref = mixed_sizes_leg(capi, reps=20, sizes=[(154, 100), (109, 164), (181, 153), (300, 194)] * 16)
        ref["us_per_clone"] = round(ref["ms_per_step"] * 1e3 / 64, 2)
        mixed_sizes["reference_patch_sizes_x16"] = {k_: ref[k_] for k_ in ref if k_ not in ("note", "streams", "group", "clones", "roi_range")}
    c5_projected = None
    if comm.rank == 0 and args.gpus == 1 and not args.no_c5_projection and args.config == "c3":
        c5_projected = c5_projection_leg(capi)
    total_pix = comm.sum(float(W * H * args.batch)) * args.steps
    value = total_pix / elapsed / 1e6
    step_traffic = profile.get("step_traffic_bytes") if profile else None
    line = {
        "metric": ("Mpix/s seamlessClone (BASELINE config 5: 64 x ROI 1024^2 sharded i mod N)" if args.config == "c5" else
                   "Mpix/s seamlessClone (ROI 2048^2)" if args.roi == 2048 else f"Mpix/s seamlessClone (ROI {args.roi}^2)"),
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if args.config == "c5" else "weak",
        "vs_baseline": None,
        "dtype": ("f32" if (opts["flags"] & capi.SC_FLAG_FLOAT_FIELD and opts["flags"] & capi.SC_FLAG_FLOAT_RHS) or args.method != "mg" else
                  "f32 (arithmetic); storage: 16-bit fixed-point field on the first two level-0 stores of a solve, float16 right-hand side and level 1"),
        "data": "synthetic",
        "data_generator": ("BatchSynth: SURVEY 8d's smooth parts + ONE pair of unit-variance noise fields per rank (seed %s), image k = smooth + sigma x the "
                           "noise rolled by a k-dependent offset, clipped to 8 bits -- the statistics of synth(roi, seed), not its exact pixels (round 4: "
                           "~0.1 s per image instead of ~0.8); image 0 of rank 0 is synth(roi, 0) itself -- the CPU baseline and the parity figures use it%s"
                           % ("3000 + rank" if image_ids is not None else "1001 + 7919 x rank",
                              "; config 5's images are NOT the per-image seeds 3000 + i of SURVEY 8d" if image_ids is not None else "")),
        "config": {"workload": (f"BASELINE config 5: 64 independent {W}x{H}-ROI clones in total, image i on rank i mod {args.gpus} (this rank: {args.batch}); "
                                if args.config == "c5" else "") +
                               f"{args.batch} independent {W}x{H}-ROI NORMAL_CLONEs per GPU per step on {streams} HIP "
                               f"streams, {group} clones per set of solver launches, 3-channel u8 images resident in HBM, solver={args.method}, "
                               + ("result = exact 5-point solution (SC_FLAG_EXACT_TABLES), +-1 grey level vs the exact-denominator port"
                                  if args.exact_tables else
                                  "result = the reference's / OpenCV's float32-table arithmetic (multigrid + float-table correction), "
                                  "+-1 grey level vs the float-table CPU port"),
                   "roi": [W, H], "dst": list(dst.shape[:2]), "batch_per_gpu": args.batch, "streams_per_gpu": streams,
                   "clones_per_launch_group": group,
                   "parallelism": f"{args.gpus} GPU(s) x {args.batch} independent images, no collective",
                   "cycles_or_sweeps": int(group_cycles)},
        "single_clone": single_clone,
        "value_without_in_step_restore": ({"value": round(comm.sum(float(W * H * args.batch)) * value_fresh[1] / value_fresh[0] / 1e6, 2), "unit": "Mpix/s",
                                           "ms_per_step": round(value_fresh[0] / value_fresh[1] * 1e3, 4), "steps": value_fresh[1],
                                           "same_bytes_as_the_restoring_step": value_fresh[2],
                                           "note": "the timed step of `value` minus its destination refresh: every step clones (in place) into its own "
                                                   "copies of the destinations, resident before the clock starts; `value` keeps the refresh inside the step"}
                                          if value_fresh else None),
        "value_float32_storage": ({"value": round(total_pix / value_f32[0] / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(value_f32[0] / args.steps * 1e3, 4),
                                   "cycles": int(value_f32[1]), "flags": "SC_FLAG_FLOAT_FIELD | SC_FLAG_FLOAT_RHS",
                                   "note": "the same timed step (same steps, same clock) with float32 storage throughout: field between the "
                                           "level-0 launches, right-hand side, level 1"} if value_f32 else None),
        "per_rank": {"elapsed_ms": per_rank_ms, "device": [int(d) for d in devices], "cores_of_rank0": len(cores), "affinity": args.affinity},
        "single_clone_stages_ms": {"mask": round(info.ms_mask, 4), "pre": round(info.ms_pre, 4), "solve": round(info.ms_solve, 4),
                                   "post": round(info.ms_post, 4), "device_total": round(info.ms_device_total, 4),
                                   "note": "one clone alone on the GPU, hipEvent marks; solve includes the float-table correction and, "
                                           "when post = 0, the post-process enqueued directly behind it"},
        "roofline": roofline, "roofline_red_black": roofline_rb, "roofline_jacobi": roofline_j,
        "roofline_jacobi_single_sweep": roofline_j1, "roofline_c4": roofline_c4, "roofline_direct_dst": roofline_dst, "roofline_direct_fft": roofline_fft,
        "source_fingerprint": src_fp,
        "whole_step_fabric": ({"bytes_per_step": int(step_traffic), "TB_per_s": round(step_traffic / (elapsed / args.steps) / 1e12, 3),
                               "frac_of_peak": round(step_traffic / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                               "traffic_source": f"{profile['_file']} @ {profile.get('git', '?')}"} if step_traffic else
                              {"bytes_per_step": None, "frac_of_peak": None, "traffic_stale": traffic_stale}),
        "pcie": pcie,
        "new_size": new_size,
        "mixed_sizes": mixed_sizes,
        "c5_projected_at_8": c5_projected,
    }
    if comm.rank == 0 and args.gpus == 1 and args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(dst, patch, mask, cx, cy, out_float, out_exact, args.cpu_seconds)
    elif comm.rank == 0:
        line["cpu_baseline"] = None
    for j in jobs:
        for key in ("f", "b0", "b", "m"):
            inst.free(j[key])
    pool.close()
    comm.barrier()
    if comm.rank == 0:
        print(json.dumps(line), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
