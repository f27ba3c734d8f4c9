python -m pytest tests -m gpu -x -q > gpurun_out/s11_tests.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/s11_tests.log | head; tail -30 gpurun_out/s11_tests.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -25
for cfg in "EBEN_DISC_ENGINE=0" "EBEN_DISC_ENGINE=1" "EBEN_DISC_ENGINE=0" "EBEN_DISC_ENGINE=1"; do
  echo "== cfg: $cfg"
  env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"
done
