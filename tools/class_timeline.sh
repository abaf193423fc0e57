#!/bin/bash
# GPU box: kernel-by-kernel timeline (both streams) of ONE size-class call of 16 members on one instance.
# usage: tools/class_timeline.sh [lo hi] -> gpurun_out/r5_class_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_c
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_c -- python3 $R/tools/group_host_probe.py --lo ${1:-300} --hi ${2:-340} --reps 5 --only ${3:-class} > /dev/null 2> $R/gpurun_out/tl_c.err || exit 1
cd $R && python3 - > gpurun_out/r5_class_timeline.txt <<'PY'
import csv, glob, re
f = sorted(glob.glob('gpurun_out/tl_c/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-24:]
t0 = int(rows[0]["Start_Timestamp"])
print("start_us  end_us  duration  queue  kernel   (the last sc_hip_run_device_batch call of tools/group_host_probe.py --only class)")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sc::", "").replace("sc::", "")
    print("%8.1f %8.1f dur %6.1f q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n[:60]))
PY
rm -rf $R/gpurun_out/tl_c
cat $R/gpurun_out/r5_class_timeline.txt
