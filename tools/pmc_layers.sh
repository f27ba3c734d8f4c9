R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/pmc1 -o p -- python $R/tools/layer_bench.py --filter melgan_discriminator.discriminator --min-gmacs 10 --iters 2 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/pmc2 -o p -- python $R/tools/layer_bench.py --filter melgan_discriminator.discriminator --min-gmacs 10 --iters 2 > $R/gpurun_out/pmc2.log 2>&1
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
