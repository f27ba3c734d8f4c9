mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu -k "phases_as_rows" > gpurun_out/r04e/tests.log 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r04e/tests.log | tail -15
python tools/layer_bench_bl.py > gpurun_out/r04e/layers_pr.txt 2>&1; grep -E "melgan|TOTAL" gpurun_out/r04e/layers_pr.txt
EBEN_DX_PR=0 python tools/layer_bench_bl.py > gpurun_out/r04e/layers_nopr.txt 2>&1; grep -E "melgan|TOTAL" gpurun_out/r04e/layers_nopr.txt
