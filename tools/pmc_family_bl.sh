# HBM traffic (TCC counters, separate rocprofv3 passes as MI355X_MICROARCH.md prescribes) of the launches the roofline records name:
# MelGAN L4 (forward = the bench's roofline kernel, weight gradient), one PQMF-band mid layer (forward).  Writes
# gpurun_out/<tag>_pmc_<layer>_{fetch,write,l2}.csv; tools/pmc_family_summary.py turns them into profiles/<tag>_pmc_*.json
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in melgan.4 pqmf0.4; do
  for C in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${C%%:*}; ctr=${C#*:}
    rocprofv3 --pmc $ctr --kernel-trace -d $O/${T}_pmc_tmp -o p -- python $R/tools/layer_bench_bl.py --only $L --iters 3 > $O/${T}_pmc_${L}_$n.log 2>&1
    python $R/tools/rocpd_pmc.py $O/${T}_pmc_tmp/p_results.db --agg --min-us 20 > $O/${T}_pmc_${L}_$n.csv; rm -rf $O/${T}_pmc_tmp
  done
done
python $R/tools/pmc_family_summary.py $T ${2:-unknown}
