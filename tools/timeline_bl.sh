# kernel trace of the bench step (plan from $1, default bf16_bl) -> per-queue Gantt and overlap timeline under gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${2:-r03}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/${T}_tr -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-leg --disc-math ${1:-bf16_bl} > $O/${T}_tr.json 2> $O/${T}_tr.err
python $R/tools/rocpd_gantt.py $O/${T}_tr/p_results.db --min-us 60 > $O/${T}_gantt.txt
python $R/tools/rocpd_timeline.py $O/${T}_tr/p_results.db --top 14 > $O/${T}_timeline.txt
python $R/tools/rocpd_gaps.py $O/${T}_tr/p_results.db > $O/${T}_gaps.txt 2>/dev/null
rm -rf $O/${T}_tr
