#!/bin/bash
# the -m gpu tests not re-run since the last full suite (the long ones -- trace pin, search, decisions, full size -- ran at this commit: gpu_call_r04s2.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04rest; mkdir -p $O; export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_trace_pin.py --deselect tests/test_search.py --deselect tests/test_decisions.py --deselect tests/test_gpu_fullsize.py 2>&1 | tail -5 > $O/pytest.log
