cd $GRAFT_REPO_ROOT; O=gpurun_out
python tools/group_host_probe.py --lo 300 --hi 340 > $O/ghp_320.json 2>&1 && python tools/group_host_probe.py --lo 120 --hi 190 > $O/ghp_150.json 2>&1 && bash tools/class_timeline.sh 300 340 > /dev/null && cp $O/r5_class_timeline.txt $O/tl_320b.txt
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tall.txt 2>&1; echo "tests rc $?"; tail -3 $O/tall.txt
