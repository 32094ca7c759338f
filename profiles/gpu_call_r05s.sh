#!/bin/bash
# GPU call r05s: havoc_mi355x_search_gate -- k_search_rows waiting, CTU row by CTU row, for reference pictures that arrive in bands on another stream
tag=${1:-r05s}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_search_gate.py -m gpu -q --durations=6 -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | head -12
grep -E "AssertionError|s call" $O/pytest.log | cut -c1-900 | head -12
