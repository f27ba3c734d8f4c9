# Kernel-trace profile of the bench command: rocprofv3's own --stats CSV + per-queue Gantt + timeline.  Usage: bash tools/profile_step.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r02}; shift; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg "$@" > $O/${T}_s.json 2> $O/${T}_s.err
f=$(find $O/${T}_s -name "*kernel_stats.csv" | head -1); cp "$f" $O/${T}_rocprofv3_kernel_stats.csv; rm -rf $O/${T}_s
rocprofv3 --kernel-trace -d $O/${T}_g -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-leg "$@" > $O/${T}_g.json 2> $O/${T}_g.err
python $R/tools/rocpd_gantt.py $O/${T}_g/p_results.db --min-us 40 > $O/${T}_gantt.txt
python $R/tools/rocpd_timeline.py $O/${T}_g/p_results.db --top 12 > $O/${T}_timeline.txt
python $R/tools/rocpd_stats.py $O/${T}_g/p_results.db --top 70 > $O/${T}_kernel_trace_stats.txt
rm -rf $O/${T}_g
