cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 300 python tools/small_call_probe.py > $O/small_call4.jsonl 2> $O/small_call.err; tail -2 $O/small_call.err
timeout -k 10 600 python -m pytest tests -q -x -m gpu -k "cli or strided or host or views or disjoint or c1 or python_class or extreme or wrong" > $O/thost.txt 2>&1; echo "tests rc $?"; tail -3 $O/thost.txt
