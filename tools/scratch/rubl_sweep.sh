set -x
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bundle_layout_residual_unit" > gpurun_out/r04b/tests.log 2>&1; tail -15 gpurun_out/r04b/tests.log
python tools/ru_bench.py > gpurun_out/r04b/ru_bench_default.txt 2>&1; cat gpurun_out/r04b/ru_bench_default.txt
for v in "EBEN_RUBL_NW32=8 EBEN_RUBL_NW64=8" "EBEN_RUBL_G128=1" "EBEN_RUBL_RT64=2" "EBEN_RUBL_BKT128=64" "EBEN_RUBL_RT128=2" "EBEN_RUBL_SEG32=1024 EBEN_RUBL_SEG64=1024 EBEN_RUBL_SEG128=1024" "EBEN_RUBL_SEG32=256 EBEN_RUBL_SEG64=256 EBEN_RUBL_SEG128=256"; do
  echo "== $v"; env $v python tools/ru_bench.py --bl-only 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04b/ru_bench_variants.txt 2>&1
cat gpurun_out/r04b/ru_bench_variants.txt
