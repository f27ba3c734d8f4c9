# Round measurement set at ONE commit (bench default = bundle-layout plan): the bench line (CPU baseline first, fp32 legs), rocprofv3
# --kernel-trace --stats of the same command, the step trace with dispatch order (per-kernel time per step, GPU busy time, the named MelGAN
# launches -> <tag>_instep_layers.{txt,json}: what bench.py's `roofline.profile` quotes), per-layer mixed-roofline table, phase and host
# times, ResidualUnit / PQMF / generator-conv / MRSTFT / feature-matching tables, generator timelines, the single-rank RCCL line, BASELINE
# config 4, counter traffic, wait classes of the roofline kernel.  Usage: bash tools/measure_round_r06.sh <tag> <commit>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r06}; C=${2:-unknown}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/${T}_st -o p -- python $R/tools/step_trace.py --steps 10 > $O/${T}_st.log 2>&1
python $R/tools/step_trace_report.py $O/${T}_st/p_results.db --layers --named $O/${T}_instep_layers.json > $O/${T}_instep_layers.txt 2>&1; rm -rf $O/${T}_st
cp $O/${T}_instep_layers.json $R/profiles/${T}_instep_layers.json   # the bench line below quotes this round's named launches
bash $R/tools/pmc_family_bl.sh $T $C; cp $O/${T}_pmc_family.json $R/profiles/${T}_pmc_family.json 2>/dev/null
python $R/bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -2 $O/${T}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/${T}_stats_bench.json 2> $O/${T}_stats.err
cp "$(find $O/${T}_s -name '*kernel_stats.csv' | head -1)" $O/${T}_rocprofv3_kernel_stats.csv; rm -rf $O/${T}_s
python $R/tools/layer_bench_bl.py --iters 20 > $O/${T}_layers_bl.txt 2>&1
EBEN_DISC_MATH=bf16_bl python $R/tools/phase_times.py > $O/${T}_phases.txt 2>&1
NO_D_UPDATE=1 EBEN_DISC_MATH=bf16_bl python $R/tools/phase_times.py > $O/${T}_phases_no_d_update.txt 2>&1
python $R/tools/host_times.py > $O/${T}_host_times.txt 2>&1
python $R/tools/ru_bench.py > $O/${T}_ru_bench.txt 2>&1
python $R/tools/pqmf_bench.py > $O/${T}_pqmf.txt 2>&1
python $R/tools/gen_conv_bench.py > $O/${T}_gen_conv.txt 2>&1
python $R/tools/mrstft_time.py > $O/${T}_mrstft.txt 2>&1
python $R/tools/fm_bench.py > $O/${T}_fm.txt 2>&1
python $R/tools/gen_fwd_timeline.py > $O/${T}_gen_fwd_timeline.txt 2>&1
python $R/tools/gen_bwd_timeline.py > $O/${T}_gen_bwd_timeline.txt 2>&1
python $R/bench.py --force-ddp --no-cpu-baseline --no-f32-leg > $O/${T}_force_ddp.json 2> $O/${T}_force_ddp.err
python $R/bench.py --workload noisybwe --no-cpu-baseline --no-f32-leg > $O/${T}_noisybwe.json 2> $O/${T}_noisybwe.err
for v in "EBEN_SPLIT_BWD=0 EBEN_SPLIT_MELGAN=0 EBEN_D_BWD_SPREAD=1" "EBEN_FIR1=0 EBEN_STFT_FRAMES_T=0 EBEN_OLA_TILED=0 EBEN_PWGEMM_FM=2" "X=1"; do
  env $v python $R/bench.py --no-cpu-baseline --no-f32-leg --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['host_enqueue_ms_per_step'])" >> $O/${T}_ablations.txt
done
bash $R/tools/l4_waits.sh ${T} melgan.4 > $O/${T}_l4_waits.log 2>&1
ls $O | grep ${T}_
