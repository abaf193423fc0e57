#!/bin/bash
# GPU box: kernel timeline of one solo clone
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/solo
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/solo -- python3 $R/tools/solo_trace.py ${1:-2048} 10 ${2:-0} ${3:-0} > /dev/null 2>&1 || exit 1
cd $R && python tools/trace_timeline.py $(ls -t gpurun_out/solo/*/*kernel_trace.csv | head -1) > gpurun_out/solo_timeline.txt; tail -1 gpurun_out/solo_timeline.txt
