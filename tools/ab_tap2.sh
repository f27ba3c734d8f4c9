timeout 900 python -m pytest tests/test_augment.py -m gpu -x -q 2>&1 | tail -12
