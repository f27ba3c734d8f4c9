R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT" "SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  rocprofv3 --pmc $C --kernel-trace -d $O/l4_pmc_tmp -o p -- python $R/tools/layer_bench_bl.py --only melgan.4 --iters 3 > $O/l4_pmc.log 2>&1
  python $R/tools/rocpd_pmc.py $O/l4_pmc_tmp/p_results.db --agg --min-us 50 2>&1 | grep -E "^kernel|tap3|bl_dw" | cut -c1-220; rm -rf $O/l4_pmc_tmp
done
tail -3 $O/l4_pmc.log
