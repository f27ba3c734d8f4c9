R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in "TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --pmc $C --kernel-trace -d $O/ru_pmc_tmp -o p -- python $R/tools/ru_bench.py --bl-only > $O/ru_pmc.log 2>&1
  python $R/tools/rocpd_pmc.py $O/ru_pmc_tmp/p_results.db --agg --min-us 15 2>&1 | grep -E "^kernel|ru3_fwd|rubl_bwd|rubl_dw" | cut -c1-200; rm -rf $O/ru_pmc_tmp
done
tail -5 $O/ru_pmc.log
