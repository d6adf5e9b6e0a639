#!/bin/bash
timeout 600 python -m pytest tests/test_dropin_gpu.py tests/test_programs_gpu.py -x -q -m gpu 2>&1 | tail -15
