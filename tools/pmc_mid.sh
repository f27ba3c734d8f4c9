R=${GRAFT_REPO_ROOT:-/root/repo}; T=r02pmc; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
F="--filter pqmf_discriminators.0 --batch 64 --math bf16 --iters 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/${T}_a -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/${T}_b -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/${T}_c -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_c.log 2>&1
for k in a b c; do python $R/tools/rocpd_pmc.py $R/gpurun_out/${T}_$k/p_results.db --match ${EBEN_PMC_MATCH:-tap3} --agg > $R/gpurun_out/${T}_$k.csv; rm -rf $R/gpurun_out/${T}_$k; done
cat $R/gpurun_out/${T}_a.csv
