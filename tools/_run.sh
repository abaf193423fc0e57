cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tall.txt 2>&1; echo "tests rc $?"; tail -4 $O/tall.txt
bash tools/mixed_profile.sh spl > /dev/null && grep -E "k_splice_planar_group|k_postprocess" $O/spl_mixed_kernel_stats.csv $O/spl_same_kernel_stats.csv | cut -c1-200
