cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/r05p_tests.txt; cat $O/r05p_tests.txt
