#!/bin/bash
# GPU box: the rocprofv3 evidence of round 5.  (The counter passes run bench.py without its float32-storage, no-restore and new-size legs, so that
# the step-traffic difference (3 steps - 1 step) / 2 is the DEFAULT step's alone; the statistics run keeps the float32 leg: its symbols
# ..., 0> / 16> / 80> / 56> are that leg's launches.)  (1) kernel statistics of `python3 bench.py`; (2) four counter passes over the same
# command (FETCH_SIZE / WRITE_SIZE x 3 steps / 1 step) -> <tag>_pmc_traffic_bench.json (copy to profiles/r5_pmc_traffic_bench.json);
# (3) tools/c4_probe.py (BASELINE config 4, 4096^2) once without the profiler for the timings (counter collection serialises the
#     launches and more than doubles their duration) and twice under it for FETCH_SIZE / WRITE_SIZE -> <tag>_c4_pmc.json (copy to
#     profiles/r5_c4_pmc.json).
# --pmc is never combined with tracing domains beyond --kernel-trace; the program comes directly after `--`; every run is under
# its own `timeout -k 10` (a GPU abort under rocprofv3 otherwise hangs until the lease is killed: round 2, s1d.err).
# usage: tools/r4_profile.sh <tag> [extra bench args]      outputs: gpurun_out/<tag>_*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-r5}; [ $# -gt 0 ] && shift
GIT=${GIT_HASH:-unknown}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_stats $O/${T}_f3 $O/${T}_w3 $O/${T}_f1 $O/${T}_w1 $O/${T}_c4f $O/${T}_c4w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $R/bench.py --cpu-seconds 0 --no-new-size --no-mixed-sizes --no-c5-projection "$@" \
    > $O/${T}_bench_under_stats.json 2> $O/${T}_stats.err || { echo "stats run failed"; tail -5 $O/${T}_stats.err; exit 1; }
cp $(ls $O/${T}_stats/*/*kernel_stats.csv | head -1) $O/${T}_kernel_stats.csv
echo "stats done"
for pass in "f3 FETCH_SIZE 3" "w3 WRITE_SIZE 3" "f1 FETCH_SIZE 1" "w1 WRITE_SIZE 1"; do
  set -- $pass
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/${T}_$1 -- python3 $R/bench.py --cpu-seconds 0 --no-c4 --no-float32-leg --no-fresh-leg --no-new-size --no-mixed-sizes --no-c5-projection --host-calls 3 --warmup 0 --steps $3 --kernel-launches 6 \
      > /dev/null 2> $O/${T}_$1.err || { echo "pmc pass $1 failed"; tail -5 $O/${T}_$1.err; exit 1; }
  echo "pass $1 done"
done
(cd $R && python3 tools/pmc_traffic_bench.py gpurun_out/${T}_f3 gpurun_out/${T}_w3 gpurun_out/${T}_f1 gpurun_out/${T}_w1 gpurun_out/${T}_pmc_traffic_bench.json 2048 32 16 $GIT)
timeout -k 10 200 python3 $R/tools/c4_probe.py 50 > $O/${T}_c4_probe.json 2> $O/${T}_c4_probe.err || { echo "c4 probe (unprofiled timings) failed"; tail -5 $O/${T}_c4_probe.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${T}_c4f -- python3 $R/tools/c4_probe.py 10 > $O/${T}_c4_probe_under_pmc.json 2> $O/${T}_c4f.err || { echo "c4 fetch pass failed"; tail -5 $O/${T}_c4f.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${T}_c4w -- python3 $R/tools/c4_probe.py 10 > /dev/null 2> $O/${T}_c4w.err || { echo "c4 write pass failed"; tail -5 $O/${T}_c4w.err; exit 1; }
(cd $R && python3 tools/c4_fold.py gpurun_out/${T}_c4f gpurun_out/${T}_c4w gpurun_out/${T}_c4_probe.json gpurun_out/${T}_c4_pmc.json $GIT)
cd $R
rm -rf gpurun_out/${T}_f3 gpurun_out/${T}_w3 gpurun_out/${T}_f1 gpurun_out/${T}_w1 gpurun_out/${T}_c4f gpurun_out/${T}_c4w
find gpurun_out/${T}_stats -name "*kernel_trace.csv" -delete
echo "profile $T done"
