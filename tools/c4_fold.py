"""Folds the two counter passes over tools/c4_probe.py into profiles/r3_c4_pmc.json.
usage: c4_fold.py <fetch dir> <write dir> <probe json line file> <out.json> <git>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_traffic import load
from seamlesscloneoptimization_amd.capi import source_fingerprint

fd, wd, probe, out, git = sys.argv[1:6]
fetch, nf = load(fd + "/**/*counter_collection.csv", "FETCH_SIZE")
write, nw = load(wd + "/**/*counter_collection.csv", "WRITE_SIZE")
timings = json.loads([l for l in open(probe) if l.startswith("{")][-1])
alg = 12 * 4094 * 4094 * 3
kern = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    kern[k] = {"launches_sampled": nf.get(k, nw.get(k, 0)), "FETCH_SIZE_KiB_avg": round(f_kib, 1), "WRITE_SIZE_KiB_avg": round(w_kib, 1),
               "traffic_bytes_per_launch": int((2 * f_kib + w_kib) * 1024)}
sweeps = {}
for name, sym in (("k_jacobi_roll<4>", "k_jacobi_roll<4, 1>"), ("k_jacobi<16>", "k_jacobi<16, 0>"), ("k_jacobi<32>", "k_jacobi<32, 0>")):
    tr = next((v["traffic_bytes_per_launch"] for k, v in kern.items() if sym in k), None)
    t = timings.get(name, {})
    us = t.get("us_per_launch")
    sweeps[name] = {"profiler_symbol": "sc::" + sym, "us_per_launch_hip_events_unprofiled": us,
                    "algorithmic_bytes_per_launch": alg, "traffic_bytes_per_launch": tr,
                    "traffic_over_algorithmic": round(tr / alg, 3) if tr else None,
                    "algorithmic_TBps": round(alg / (us * 1e-6) / 1e12, 3) if us else None,
                    "frac_of_8TBps_algorithmic": round(alg / (us * 1e-6) / 8e12, 4) if us else None,
                    "frac_of_8TBps_counter_traffic": round(tr / (us * 1e-6) / 8e12, 4) if us and tr else None}
json.dump({"note": "BASELINE config 4: 4096^2 ROI, 3 channels, one Jacobi sweep per launch; bytes crossing the L2 -> fabric boundary per "
                   "launch = 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE, KiB units, separate rocprofv3 --pmc passes over "
                   "tools/c4_probe.py (MI355X_MICROARCH.md, HBM section); timings are HIP-event means of an UNPROFILED run of the same probe "
                   "on the same box (under counter collection a launch takes 2.5x as long: never price bytes from one run with time "
                   "from the other kind)",
           "roi": 4096, "git": git, "source_fingerprint": source_fingerprint(), "probe": timings, "single_sweep_kernels": sweeps,
           "kernels": kern}, open(out, "w"), indent=1)
for n, v in sweeps.items():
    print(n, v)
