#!/bin/bash
# round 3, GPU call J: smoke, the whole GPU suite (with the reference encoder over the MI355X tables), then the round's profile collection
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -3 $O/smoke.log
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
bash profiles/collect.sh r03 > $O/collect.log 2>&1
tail -2 $O/collect.log | cut -c1-300
