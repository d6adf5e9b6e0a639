timeout 600 python -m pytest tests/test_query_gpu.py -x -q -m gpu -k weight_test_is_elided 2>&1 | tail -40
