mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu > gpurun_out/r04f/tests_bl.log 2>&1; grep -E "passed|failed" gpurun_out/r04f/tests_bl.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "bundle_layout or benchmarked or full_size or replayed or bitwise or trajectory or tolerances or BF16 or bf16" > gpurun_out/r04f/tests_models.log 2>&1; grep -E "passed|failed" gpurun_out/r04f/tests_models.log | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err; tail -1 gpurun_out/r04f/bench.err
EBEN_DX_PR=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04f/bench_nopr.json 2> gpurun_out/r04f/bench_nopr.err; tail -1 gpurun_out/r04f/bench_nopr.err
