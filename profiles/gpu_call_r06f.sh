#!/bin/bash
# GPU call r06f: the reference encoder over the drop-in tables on the MI355X (hooked: pictures registered) -- stream identical, seconds, launches
tag=${1:-r06f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_reference_encoder.py -m gpu -q -x -s -p no:cacheprovider -k "hooked" > $O/pytest_hooked.log 2>&1; echo "hooked: $(tail -1 $O/pytest_hooked.log)"; grep -E "^gpu_ra|^E |^FAILED" $O/pytest_hooked.log | cut -c1-400 | head -8
