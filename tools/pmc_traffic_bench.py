"""Folds the rocprofv3 --pmc passes of `bench.py` itself (tools/r2_profile.sh) into profiles/r2_pmc_traffic_bench.json:
fabric-side bytes per launch of every kernel (2 x FETCH_SIZE + WRITE_SIZE: gfx950 half-count correction on the read side,
KiB units -- MI355X_MICROARCH.md, HBM section; FETCH_SIZE and WRITE_SIZE come from separate passes) and the bytes of one
timed step (difference of the totals of a 3-step and a 1-step run, divided by 2).

usage: pmc_traffic_bench.py <fetch dir, 3 steps> <write dir, 3 steps> <fetch dir, 1 step> <write dir, 1 step> <out.json> roi batch group git"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seamlesscloneoptimization_amd.capi import source_fingerprint   # hashes source files only: no library is loaded


def load(d, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    total = 0.0
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            v = float(row["Counter_Value"])
            a = per[row["Kernel_Name"]]; a[0] += v; a[1] += 1
            total += v
    return per, total


f3, w3, f1, w1, out = sys.argv[1:6]
roi, batch, group, git = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
pf, tf3 = load(f3, "FETCH_SIZE"); pw, tw3 = load(w3, "WRITE_SIZE")
_, tf1 = load(f1, "FETCH_SIZE"); _, tw1 = load(w1, "WRITE_SIZE")
kern = {}
for k in sorted(set(pf) | set(pw)):
    fk = pf[k][0] / max(pf[k][1], 1) if k in pf else 0.0
    wk = pw[k][0] / max(pw[k][1], 1) if k in pw else 0.0
    kern[k] = {"launches_sampled": pf[k][1] if k in pf else pw[k][1], "FETCH_SIZE_KiB_avg": round(fk, 1), "WRITE_SIZE_KiB_avg": round(wk, 1),
               "traffic_bytes_per_launch": int((2 * fk + wk) * 1024)}
step = ((2 * tf3 + tw3) - (2 * tf1 + tw1)) * 1024 / 2.0
json.dump({"note": "bytes crossing the L2 -> fabric boundary (Infinity-Cache hits included); read side = 2 x FETCH_SIZE (gfx950 "
                   "correction), KiB units; separate rocprofv3 --pmc passes over `python3 bench.py --cpu-seconds 0 --warmup 0 --steps {3,1}`",
           "roi": roi, "batch": batch, "group": group, "git": git, "source_fingerprint": source_fingerprint(), "step_traffic_bytes": int(step), "kernels": kern}, open(out, "w"), indent=1)
print("step traffic %.2f GB" % (step / 1e9))
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:14]:
    print(f"{v['traffic_bytes_per_launch']/1e6:10.2f} MB x{v['launches_sampled']:5d}  {k[:100]}")
