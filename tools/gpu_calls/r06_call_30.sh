#!/bin/bash
timeout 600 python -m pytest tests/test_query_gpu.py -x -q -m gpu -k "skips_what" 2>&1 | tail -30
