mkdir -p gpurun_out/r04c
timeout 1400 python -m pytest tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r04c/tests_models.log 2>&1; tail -5 gpurun_out/r04c/tests_models.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; tail -1 gpurun_out/r04c/bench.err
EBEN_RU_BL=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04c/bench_nobl.json 2> gpurun_out/r04c/bench_nobl.err; tail -1 gpurun_out/r04c/bench_nobl.err
python tools/phase_times.py > gpurun_out/r04c/phases.txt 2>&1; cat gpurun_out/r04c/phases.txt
