# SQ counters of the heavy discriminator layers (separate passes: 8 SQ slots each).  Usage: pmc_layers.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-pmc}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
F="--filter melgan_discriminator.discriminator --min-gmacs 10 --iters 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/${T}_a -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/${T}_b -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/${T}_c -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/${T}_c.log 2>&1
ls $R/gpurun_out/${T}_a $R/gpurun_out/${T}_b $R/gpurun_out/${T}_c
