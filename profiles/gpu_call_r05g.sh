#!/bin/bash
# GPU call r05g: the whole -m gpu suite at this commit (log kept), smoke() as the driver calls it, then the round's profiles and bench line (profiles/collect.sh)
tag=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "suite: $(tail -1 $O/pytest_gpu.log)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash profiles/collect.sh r05 > $O/collect.log 2>&1
tail -1 $O/collect.log | cut -c1-600
