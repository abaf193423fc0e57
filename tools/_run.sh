cd $GRAFT_REPO_ROOT; O=gpurun_out
: > $O/mp4.jsonl
for r in "1000 1100" "300 340" "120 190" "2000 2200"; do set -- $r; timeout -k 10 200 python tools/mixed_probe.py --lo $1 --hi $2 --legs mixed >> $O/mp4.jsonl || exit 1; done
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tall.txt 2>&1; echo "tests rc $?"; tail -3 $O/tall.txt
