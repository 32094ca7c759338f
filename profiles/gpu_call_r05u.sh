#!/bin/bash
# GPU call r05u: DecisionPicture.step_banded -- the producer's half of CTU-row bands: everything after the searches band by band on a second stream behind
# havoc_mi355x_search_wait_rows, while the rows below are searched; must leave what step() leaves
tag=${1:-r05u}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_step_banded.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED|whole step" $O/pytest.log | cut -c1-400 | head -20
