mkdir -p gpurun_out/r04i
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu > gpurun_out/r04i/tests_bl.log 2>&1; grep -E "passed|failed" gpurun_out/r04i/tests_bl.log | tail -3
python tools/layer_bench_bl.py > gpurun_out/r04i/layers.txt 2>&1; grep -E "head|tail|TOTAL" gpurun_out/r04i/layers.txt
EBEN_BL_HEAD_DX4=0 EBEN_BL_TAIL_PAIRS=8 python tools/layer_bench_bl.py 2>&1 | grep -E "head|tail|TOTAL"
