R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/tools/gen_conv_bench.py
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  rocprofv3 --pmc $C --kernel-trace -d $O/gc_pmc_tmp -o p -- python $R/tools/gen_conv_bench.py --only enc3 --iters 3 > $O/gc_pmc.log 2>&1
  python $R/tools/rocpd_pmc.py $O/gc_pmc_tmp/p_results.db --agg --min-us 20 2>&1 | grep -E "gc_kernel|counter|name" | head -8; rm -rf $O/gc_pmc_tmp
done
tail -3 $O/gc_pmc.log
