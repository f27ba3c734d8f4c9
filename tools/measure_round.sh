# Round measurement set: bench line (CPU baseline, fp32 leg), rocprofv3 --kernel-trace --stats of the same command, the three PMC
# traffic passes of the roofline kernel + their summary, per-queue Gantt / overlap timeline, per-layer tables, phase times.
# Usage: bash tools/measure_round.sh <tag> [commit]     (writes gpurun_out/<tag>_*; copy what is to be judged into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r02}; C=${2:-unknown}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -2 $O/${T}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/${T}_stats_bench.json 2> $O/${T}_stats.err
cp "$(find $O/${T}_s -name '*kernel_stats.csv' | head -1)" $O/${T}_rocprofv3_kernel_stats.csv; rm -rf $O/${T}_s
rocprofv3 --kernel-trace -d $O/${T}_trace -o p -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/${T}_trace.json 2> $O/${T}_trace.err
python $R/tools/rocpd_gantt.py $O/${T}_trace/p_results.db --min-us 60 > $O/${T}_gantt.txt
python $R/tools/rocpd_timeline.py $O/${T}_trace/p_results.db --top 12 > $O/${T}_timeline.txt; head -3 $O/${T}_timeline.txt
rm -rf $O/${T}_trace
bash $R/tools/pmc_traffic.sh > /dev/null 2>&1
python $R/tools/pmc_summary.py $T bf16 64 $C
python $R/tools/layer_bench.py > $O/${T}_layers.txt 2>&1
python $R/tools/layer_bench.py --batch 64 --filter D. --math bf16 > $O/${T}_layers_bf16.txt 2>&1
python $R/tools/layer_bench.py --batch 64 --filter D. --math bf16x6 > $O/${T}_layers_bf16x6.txt 2>&1
python $R/tools/ru_bench.py > $O/${T}_ru_bench.txt 2>&1
python $R/tools/phase_times.py > $O/${T}_phases.txt 2>&1
python $R/tools/plan_parity.py > $O/${T}_plans.txt 2>&1
$R/tools/ubench/mfma_bf16 > $O/${T}_mfma_bf16.txt 2>&1
ls $O | grep ${T}_
