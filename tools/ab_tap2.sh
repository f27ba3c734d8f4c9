for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done
python bench.py 2>&1 | tail -1 | cut -c1-250
