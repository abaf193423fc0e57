#!/bin/bash
# GPU box: where the waves of the bench's kernels spend their cycles -- one rocprofv3 --pmc pass with the eight SQ slots
# (MI355X_MICROARCH.md, PMC slots: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing;
# the three add up to WAVE_CYCLES).  --pmc is combined with --kernel-trace only; the program comes directly after `--`.
# usage: tools/sq_counters.sh [tag]      output: gpurun_out/<tag>_sq_counters.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-r3}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf $O/${T}_sq
timeout -k 10 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
    --output-format csv -d $O/${T}_sq -- python3 $R/bench.py --cpu-seconds 0 --no-c4 --no-new-size --no-c5-projection --no-mixed-sizes --warmup 0 --steps 1 --kernel-launches 6 > /dev/null 2> $O/${T}_sq.err || { echo "sq pass failed"; tail -5 $O/${T}_sq.err; exit 1; }
cd $R && python3 - "$O/${T}_sq" "$O/${T}_sq_counters.json" <<'PY'
import collections, csv, glob, json, sys
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        a = per[row["Kernel_Name"]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
out = {}
for k, c in per.items():
    wc = c["SQ_WAVE_CYCLES"][0]
    if wc <= 0: continue
    out[k] = {"launches": c["SQ_WAVE_CYCLES"][1], "wave_cycles_per_launch": wc / c["SQ_WAVE_CYCLES"][1]}
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        out[k][n + "_over_WAVE_CYCLES"] = round(c[n][0] / wc, 4)
    out[k]["SQ_BUSY_CYCLES_per_launch"] = c["SQ_BUSY_CYCLES"][0] / max(c["SQ_BUSY_CYCLES"][1], 1)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles_per_launch"] * kv[1]["launches"])[:12]:
    print(k[:70], {a[3:].replace("_over_WAVE_CYCLES", ""): b for a, b in v.items() if a.endswith("WAVE_CYCLES")})
PY
rm -rf $O/${T}_sq
