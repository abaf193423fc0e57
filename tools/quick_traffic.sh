#!/bin/bash
# GPU box: fabric traffic per launch of the isolated hot kernels (two separate counter passes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/qt_f $R/gpurun_out/qt_w
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/qt_f -- python3 $R/tools/pmc_probe.py ${1:-2048} > /dev/null 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/qt_w -- python3 $R/tools/pmc_probe.py ${1:-2048} > /dev/null 2>&1 || exit 1
cd $R && python tools/pmc_traffic.py gpurun_out/qt_f gpurun_out/qt_w gpurun_out/qt_traffic.json | grep -E "1>|8, 2>|2>\(" 
