timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head -5
bash tools/measure_round.sh r01g
