#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_trace_pin.py tests/test_search.py tests/test_decisions.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4 > $O/pytest.log
