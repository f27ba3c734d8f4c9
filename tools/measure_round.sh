# Round measurement set: bench line (with CPU baseline), rocprofv3 kernel trace of the same command, PMC traffic
# passes of the roofline kernel, per-layer table.  Usage: bash tools/measure_round.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r01}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -2 $O/${T}_bench.err
rocprofv3 --kernel-trace -d $O/${T}_trace -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/${T}_trace.json 2> $O/${T}_trace.err
python $R/tools/rocpd_stats.py $O/${T}_trace/p_results.db --top 60 --csv $O/${T}_kernel_trace_stats.csv > $O/${T}_kernel_trace_stats.txt
bash $R/tools/pmc_traffic.sh > /dev/null 2>&1
for k in fetch write l2; do python $R/tools/rocpd_pmc.py $O/pmc_$k/p_results.db --match ${EBEN_PMC_KERNEL:-tap3} --agg > $O/${T}_pmc_$k.csv; done
python $R/tools/layer_bench.py > $O/${T}_layers.txt 2>&1
python $R/tools/layer_bench.py --batch 64 --filter D. --math bf16 > $O/${T}_layers_bf16.txt 2>&1
rm -rf $O/${T}_trace
ls $O | grep ${T}_
