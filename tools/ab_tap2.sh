# Scratch A/B harness used during development (edit freely): GPU parity tests, then three benchmark runs.
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head -5
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
