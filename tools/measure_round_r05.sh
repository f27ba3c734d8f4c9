# Round measurement set (bench default = bundle-layout plan): bench line (CPU baseline first, fp32 legs), rocprofv3 --kernel-trace --stats of the
# same command, overlap timeline, per-layer mixed-roofline table, phase times, ResidualUnit table, PQMF kernels, generator timelines, host
# enqueue times, the single-rank RCCL line, BASELINE config 4, counter traffic.  Usage: bash tools/measure_round_r05.sh <tag> <commit>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05}; C=${2:-unknown}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -2 $O/${T}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/${T}_stats_bench.json 2> $O/${T}_stats.err
cp "$(find $O/${T}_s -name '*kernel_stats.csv' | head -1)" $O/${T}_rocprofv3_kernel_stats.csv; rm -rf $O/${T}_s
bash $R/tools/timeline_bl.sh bf16_bl $T
python $R/tools/layer_bench_bl.py --iters 20 > $O/${T}_layers_bl.txt 2>&1
EBEN_DISC_MATH=bf16_bl python $R/tools/phase_times.py > $O/${T}_phases.txt 2>&1
python $R/tools/ru_bench.py > $O/${T}_ru_bench.txt 2>&1
python $R/tools/pqmf_bench.py > $O/${T}_pqmf.txt 2>&1
python $R/tools/gen_fwd_timeline.py > $O/${T}_gen_fwd_timeline.txt 2>&1
python $R/tools/gen_bwd_timeline.py > $O/${T}_gen_bwd_timeline.txt 2>&1
python $R/tools/host_times.py > $O/${T}_host_times.txt 2>&1
python $R/bench.py --force-ddp --no-cpu-baseline --no-f32-leg > $O/${T}_force_ddp.json 2> $O/${T}_force_ddp.err
python $R/bench.py --workload noisybwe --no-cpu-baseline --no-f32-leg > $O/${T}_noisybwe.json 2> $O/${T}_noisybwe.err
bash $R/tools/pmc_family_bl.sh $T $C
# the generator's strided / transposed / latent convs stand-alone, and what occupies the 128-channel stride-8 launch (texture addresser busy
# time against the launch's cycles, instruction mix): separate --pmc passes, rows = kernel, columns named in the header line
python $R/tools/gen_conv_bench.py > $O/${T}_gen_conv.txt 2>&1
for CTRS in "TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --pmc $CTRS --kernel-trace -d $O/${T}_gcp -o p -- python $R/tools/gen_conv_bench.py --only enc3 --iters 3 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $O/${T}_gcp/p_results.db --agg --min-us 20 2>&1 | grep -E "^kernel|gc_kernel" >> $O/${T}_gen_conv.txt; rm -rf $O/${T}_gcp
done
python $R/tools/hbm_kernels.py $O/${T}_rocprofv3_kernel_stats.csv > $O/${T}_hbm_kernels.txt 2>&1
ls $O | grep ${T}_
bash $R/tools/l4_waits.sh ${T} melgan.4 > $O/${T}_l4_waits.log 2>&1
ls $O | grep ${T}_
