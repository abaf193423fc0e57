#!/bin/bash
# GPU box: kernel statistics of tools/mixed_probe.py's legs under rocprofv3 (--kernel-trace --stats only), one run per leg.
# usage: tools/mixed_profile.sh <tag> [probe args]      outputs: gpurun_out/<tag>_{mixed,same}_kernel_stats.csv, _probe.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=${1:-mix}; [ $# -gt 0 ] && shift
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for leg in mixed same; do
  rm -rf $O/${T}_${leg}_stats
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${leg}_stats -- python3 $R/tools/mixed_probe.py --legs $leg --reps 10 "$@" \
      > $O/${T}_${leg}_probe.json 2> $O/${T}_${leg}.err || { echo "$leg run failed"; tail -5 $O/${T}_${leg}.err; exit 1; }
  cp $(ls $O/${T}_${leg}_stats/*/*kernel_stats.csv | head -1) $O/${T}_${leg}_kernel_stats.csv
  rm -rf $O/${T}_${leg}_stats
  echo "$leg done"; cat $O/${T}_${leg}_probe.json
done
