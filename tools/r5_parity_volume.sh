#!/bin/bash
# GPU box: the long parity runs of tests/tools/ on the final tree (not part of the test suite) -> gpurun_out/r5_parity_volume.txt
# (every tool writes its own log under gpurun_out/ as it goes: a quiet command is taken to be hung after 7 minutes)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/r5_parity_volume.txt
: > $O
run() { # title, log, command...
    local title=$1 log=gpurun_out/$2; shift 2
    echo "== $title" >> $O
    timeout -k 10 1000 "$@" > $log 2>&1 || echo "(exit code $?)" >> $O
    if [ "$KEEP_ALL" = 1 ]; then cat $log >> $O; else tail -1 $log >> $O; fi
}
run "fuzz_classes 40 (seed 5)" pv_classes_a.log python tests/tools/fuzz_classes.py 40 5
run "fuzz_classes 30 (seed 23)" pv_classes_b.log python tests/tools/fuzz_classes.py 30 23
run "fuzz_shapes 600 (seed 21)" pv_shapes.log python tests/tools/fuzz_shapes.py 600 21
run "fuzz_mg 160 (seed 33)" pv_mg.log python tests/tools/fuzz_mg.py 160 33
run "fuzz_groups 40 big (seed 13)" pv_groups.log python tests/tools/fuzz_groups.py 40 13 big
run "soak 20 rounds x 24 jobs x 4 streams" pv_soak.log python tests/tools/soak.py 20 24 4
KEEP_ALL=1 run "large_roi_check 5000x5000 8192x8192 12000x7000" pv_large.log python tests/tools/large_roi_check.py 5000x5000 8192x8192 12000x7000
cat $O
