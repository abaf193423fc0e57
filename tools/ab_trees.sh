#!/bin/bash
# Same-box A/B of two TREES: builds another revision of this repository in a scratch subdirectory HERE (it travels to the GPU box
# with the snapshot), then alternates `bench.py --steps 20 --warmup 5` runs of both on the box.  Boxes differ by +-2 %, consecutive
# runs on one box by +-0.3 %: this is how round 5 found what it had cost the single clone and the batch step since round 4.
#   here:        tools/ab_trees.sh build <git-rev>          (e.g. 7684aa0 = round 4's final tree)
#   on the box:  gpurun -- 'bash tools/ab_trees.sh run [repetitions]'   -> gpurun_out/ab_trees.txt
#   afterwards:  tools/ab_trees.sh clean
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
case "$1" in
build)
  rm -rf _ab_old && mkdir _ab_old && git archive "$2" | tar -x -C _ab_old && make -C _ab_old/seamlesscloneoptimization_amd/csrc -j8 > /dev/null && echo "built $2 in _ab_old" ;;
run)
  O=$R/gpurun_out/ab_trees.txt; : > $O
  pick='import sys,json; b=json.loads(sys.stdin.read()); print(sys.argv[1], b["value"], b["ms_per_step"], b["single_clone"]["ms"], b["pcie"]["call_ms"])'
  for rep in $(seq 1 ${2:-3}); do
    (cd _ab_old && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-new-size --no-c4 2>/dev/null | python -c "$pick" other) >> $O
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-new-size --no-c4 --no-mixed-sizes --no-c5-projection 2>/dev/null | python -c "$pick" this >> $O
  done
  echo "tree value_Mpix/s ms_per_step single_clone_ms host_call_ms"; cat $O ;;
clean) rm -rf _ab_old ;;
*) echo "usage: $0 build <rev> | run [reps] | clean"; exit 2 ;;
esac
