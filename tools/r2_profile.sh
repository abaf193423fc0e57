#!/bin/bash
# GPU box: the rocprofv3 evidence of `python3 bench.py` -- kernel statistics of the full command, then four counter passes
# (FETCH_SIZE / WRITE_SIZE x 3 steps / 1 step; --pmc never together with --stats' tracing domains beyond --kernel-trace).
# usage: tools/r2_profile.sh <tag> [extra bench args]      outputs: gpurun_out/<tag>_*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-r2}; [ $# -gt 0 ] && shift
GIT=${GIT_HASH:-unknown}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_f3 $R/gpurun_out/${T}_w3 $R/gpurun_out/${T}_f1 $R/gpurun_out/${T}_w1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --cpu-seconds 0 "$@" \
    > $R/gpurun_out/${T}_bench_under_stats.json 2> $R/gpurun_out/${T}_stats.err || { echo "stats run failed"; tail -5 $R/gpurun_out/${T}_stats.err; exit 1; }
cp $(ls $R/gpurun_out/${T}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_kernel_stats.csv
echo "stats done"
for pass in "f3 FETCH_SIZE 3" "w3 WRITE_SIZE 3" "f1 FETCH_SIZE 1" "w1 WRITE_SIZE 1"; do
  set -- $pass
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/${T}_$1 -- python3 $R/bench.py --cpu-seconds 0 --warmup 0 --steps $3 --kernel-launches 6 \
      > /dev/null 2> $R/gpurun_out/${T}_$1.err || { echo "pmc pass $1 failed"; tail -5 $R/gpurun_out/${T}_$1.err; exit 1; }
  echo "pass $1 done"
done
cd $R && python3 tools/pmc_traffic_bench.py gpurun_out/${T}_f3 gpurun_out/${T}_w3 gpurun_out/${T}_f1 gpurun_out/${T}_w1 gpurun_out/${T}_pmc_traffic_bench.json 2048 32 16 $GIT
rm -rf gpurun_out/${T}_f3 gpurun_out/${T}_w3 gpurun_out/${T}_f1 gpurun_out/${T}_w1
find gpurun_out/${T}_stats -name "*kernel_trace.csv" -delete
