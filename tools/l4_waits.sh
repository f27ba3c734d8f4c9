# Per-wait-class PMC split of one layer's launches (default MelGAN L4: the roofline kernel), separate rocprofv3 --pmc passes of
# <= 8 SQ counters each; also records whether this box can decode a thread trace (rocprofv3 --att).
# Usage: bash tools/l4_waits.sh <tag> [layer]   -> gpurun_out/<tag>_l4_waits.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05}; L=${2:-melgan.4}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
OUT=$O/${T}_l4_waits.txt; : > $OUT
echo "# thread-trace decoder libraries on this box:" >> $OUT
find / \( -name "*trace_decoder*" -o -name "*att_decoder*" -o -name "*attdecoder*" \) 2>/dev/null | head >> $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $O/${T}_sq_counters.txt
echo "# $(wc -l < $O/${T}_sq_counters.txt) SQ counters listed by rocprofv3 -L" >> $OUT
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
            "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
            "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum"; do
  rocprofv3 --pmc $CTRS --kernel-trace -d $O/${T}_w -o p -- python $R/tools/layer_bench_bl.py --only $L --iters 3 > $O/${T}_w.log 2>&1
  python $R/tools/rocpd_pmc.py $O/${T}_w/p_results.db --agg --min-us 20 >> $OUT 2>&1; rm -rf $O/${T}_w
done
# thread trace attempt (one dispatch of the forward kernel); output kept only as a listing
timeout 300 rocprofv3 --att --kernel-trace --kernel-include-regex "tap3_kernel|big_kernel" -d $O/${T}_att -o p -- python $R/tools/layer_bench_bl.py --only $L --iters 1 > $O/${T}_att.log 2>&1
echo "# rocprofv3 --att exit $? ; files:" >> $OUT; find $O/${T}_att -type f 2>/dev/null | head -20 >> $OUT; tail -5 $O/${T}_att.log >> $OUT
du -sh $O/${T}_att 2>/dev/null >> $OUT; rm -rf $O/${T}_att
cat $OUT
