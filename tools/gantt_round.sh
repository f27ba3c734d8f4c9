R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/s61 -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/s61.json 2> $O/s61.err
python $R/tools/rocpd_gantt.py $O/s61/p_results.db --min-us 60 > $O/r01_gantt.txt; tail -1 $O/r01_gantt.txt | cut -c1-200
python $R/tools/rocpd_timeline.py $O/s61/p_results.db --top 12 > $O/r01_timeline.txt; head -3 $O/r01_timeline.txt
rm -rf $O/s61
