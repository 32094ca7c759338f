#!/bin/bash
# GPU call r06aa: an intra picture's chain with an issuing thread + stream per partition size: parity (3 sizes) and seconds per picture, against one thread (HAVOC_INTRA_CHAIN_THREADS=0)
tag=${1:-r06aa}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_intra_chain.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "threads: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED|seconds per picture" $O/pytest.log | cut -c1-300 | head -12
HAVOC_INTRA_CHAIN_THREADS=0 timeout 900 python -m pytest tests/test_intra_chain.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest_one.log 2>&1; echo "one thread: $(tail -1 $O/pytest_one.log)"; grep -E "^E |^FAILED|seconds per picture" $O/pytest_one.log | cut -c1-300 | head -12
