#!/bin/bash
# GPU call r06t: the intra picture's chain with a level's sizes side by side + one-kernel RDOQ for small launches: parity (3 sizes) and seconds per picture; RDOQ tests; decisions
tag=${1:-r06t}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_intra_chain.py tests/test_rdoq.py tests/test_decisions.py tests/test_search.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED|seconds per picture" $O/pytest.log | cut -c1-300 | head -12
