#!/bin/bash
# round 4, call r: the window sad4 as the default: its tests, the metric tests, and the bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04r; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sad4_window.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_search.py -q -m gpu 2>&1 | tail -3 > $O/pytest.log
true
