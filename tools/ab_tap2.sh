timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "bf16 disc|passed|failed|^E  " | head -8
