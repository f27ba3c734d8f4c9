R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -2
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04ad_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > /dev/null 2> /dev/null; grep -E "bl_fm_partial" "$(find $O/r04ad_s -name "*kernel_stats.csv" | head -1)"; rm -rf $O/r04ad_s
cd $R; for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done
