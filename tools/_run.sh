cd $GRAFT_REPO_ROOT; O=gpurun_out
: > $O/mp3.jsonl
for r in "1000 1100" "300 340" "120 190" "500 560" "2000 2200" "100 2400" "520 1020"; do set -- $r; timeout -k 10 200 python tools/mixed_probe.py --lo $1 --hi $2 >> $O/mp3.jsonl || exit 1; done
python tools/solo_trace.py > $O/solo_trace.txt 2>&1 || true
