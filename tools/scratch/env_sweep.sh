mkdir -p gpurun_out/r03e
for kv in "base=1" "EBEN_TAP3_X3_LDS_KB=32" "EBEN_TAP3_X3_LDS_KB=44" "EBEN_TAP3_X3_LDS_KB=52" "EBEN_TAP3_LDS_KB=24" "EBEN_TAP3_LDS_KB=32" "EBEN_TAP3_LDS_KB=64" "EBEN_TAP3_BIG_LDS_KB=52" "EBEN_TAP3_BIG_LDS_KB=64"; do
  env $kv python tools/layer_bench_bl.py --iters 20 > gpurun_out/r03e/$kv.txt 2>&1
  echo "$kv: $(tail -1 gpurun_out/r03e/$kv.txt)"
done
