#!/bin/bash
# GPU call r06x: the round's profiles at the final kernels (profiles/collect.sh r06) + the whole -m gpu suite
tag=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
bash profiles/collect.sh r06 > $O/collect.log 2>&1; tail -1 $O/collect.log | cut -c1-400
cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "gpu suite: $(tail -1 $O/pytest_gpu.log)"; grep -E "^E |^FAILED" $O/pytest_gpu.log | cut -c1-400 | head -12
