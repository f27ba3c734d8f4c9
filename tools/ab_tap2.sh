python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
F="--filter melgan_discriminator.discriminator --min-gmacs 10 --iters 5"
for cfg in "" ; do
  echo "== cfg: $cfg"
  env $cfg python tools/layer_bench.py $F 2>&1 | grep melgan | cut -c40-130
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"
