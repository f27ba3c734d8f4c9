for cfg in "EBEN_DW2=0" "EBEN_DW2=1" "EBEN_DW2=0" "EBEN_DW2=1"; do
  echo "== cfg: $cfg"
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"
done
