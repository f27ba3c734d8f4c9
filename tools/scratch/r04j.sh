mkdir -p gpurun_out/r04j
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bl.py -x -q -m gpu -n 4 > gpurun_out/r04j/tests_ops.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r04j/tests_ops.log | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04j/bench.json 2> gpurun_out/r04j/bench.err; tail -1 gpurun_out/r04j/bench.err
python tools/phase_times.py 2>&1 | grep -v amdgpu > gpurun_out/r04j/phases.txt; cat gpurun_out/r04j/phases.txt
