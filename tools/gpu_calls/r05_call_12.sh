#!/bin/bash
O=gpurun_out/r05_c12; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40) | tee $O/pytest_full.txt
