cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tall.txt 2>&1; echo "tests rc $?"; tail -2 $O/tall.txt
timeout -k 10 200 python tools/small_call_probe.py > $O/small_call7.jsonl 2>/dev/null
python tools/solo_trace.py 2048 10 0 0 2>&1 | tail -2
bash tools/quick_solo.sh 2048 > /dev/null; head -3 $O/solo_timeline.txt; tail -1 $O/solo_timeline.txt
