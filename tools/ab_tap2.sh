timeout 900 python -m pytest tests/test_reference_suite.py -m gpu -x -q 2>&1 | tail -8
