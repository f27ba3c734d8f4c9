# PMC passes (rocprofv3, counters only + kernel trace) over the bundle-layout layer bench; one CSV per pass under gpurun_out/<tag>_pmc_{a,b,c}.csv
# Usage: bash tools/pmc_bl.sh <tag> [layer_bench_bl.py filter] [kernel-name match]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03}; FIL=${2:-melgan}; M=${3:-}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
F="--filter $FIL --iters 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/${T}_pa -o p -- python $R/tools/layer_bench_bl.py $F > $R/gpurun_out/${T}_pa.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/${T}_pb -o p -- python $R/tools/layer_bench_bl.py $F > $R/gpurun_out/${T}_pb.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace -d $R/gpurun_out/${T}_pc -o p -- python $R/tools/layer_bench_bl.py $F > $R/gpurun_out/${T}_pc.log 2>&1
for k in a b c; do python $R/tools/rocpd_pmc.py $R/gpurun_out/${T}_p$k/p_results.db --match "$M" --agg --min-us 15 > $R/gpurun_out/${T}_pmc_$k.csv; rm -rf $R/gpurun_out/${T}_p$k; done
