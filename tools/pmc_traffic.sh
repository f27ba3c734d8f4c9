# HBM traffic of the dominant kernel (MelGAN L4 forward) from the TCC counters, in SEPARATE passes
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
F="--filter melgan_discriminator.discriminator.4 --iters 3 --batch 64 --math ${EBEN_PMC_MATH:-bf16}"   # the engine launches enhanced + reference together
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/pmc_l2 -o p -- python $R/tools/layer_bench.py $F > $R/gpurun_out/pmc_l2.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_l2
