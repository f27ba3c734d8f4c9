# ablations of the ResidualUnit forward (scratch builds; results wrong by construction)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04h
for bits in 1 2 3 4 8 15; do
  echo "== EBEN_RU_DBG=$bits"; EBEN_HIP_LIB=$R/variants/dbg$bits/libeben_hip.so python $R/tools/ru_bench.py --bl-only 2>&1 | grep -E "^C|sum"
done > $R/gpurun_out/r04h/ru_fwd_ablations.txt 2>&1
cat $R/gpurun_out/r04h/ru_fwd_ablations.txt
