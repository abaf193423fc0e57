set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
S=$(date +%s)
timeout -k 10 500 python bench.py > $O/fin_bench.json 2> $O/fin_bench.err && echo "bench $(( $(date +%s) - S )) s" &&
timeout -k 10 300 python bench.py --config c5 > $O/fin_bench_c5.json 2>> $O/fin_bench.err && echo c5 done &&
timeout -k 10 400 python tools/bench_configs.py > $O/fin_configs.jsonl 2>> $O/fin_bench.err && echo configs done &&
( : > $O/fin_mixed.jsonl; for r in "1000 1100" "300 340" "120 190" "500 560" "2000 2200" "520 1020" "100 2400"; do set -- $r; timeout -k 10 200 python tools/mixed_probe.py --lo $1 --hi $2 --group 0 >> $O/fin_mixed.jsonl || exit 1; done ) && echo mixed done &&
bash tools/class_timeline.sh 300 340 > /dev/null && echo timeline done &&
bash tools/mixed_profile.sh finmix --group 0 > /dev/null && echo mixprof done &&
timeout -k 10 200 python bench.py --reference-table > $O/fin_reftable.jsonl 2>> $O/fin_bench.err && echo reftable done &&
timeout -k 10 200 python tools/small_call_probe.py > $O/fin_small_call.jsonl 2>> $O/fin_bench.err && echo small done &&
bash tools/quick_solo.sh 2048 > /dev/null && cp $O/solo_timeline.txt $O/fin_solo_timeline.txt && echo solo done
