#!/bin/bash
# GPU call r05t (the round's last): the whole -m gpu suite at the final commit (log kept), smoke() as the driver calls it, the round's profiles and bench line
# (profiles/collect.sh r05), and one sequence on 8 virtual ranks re-measured with the reference's predictor derivation in the walk
tag=${1:-r05t}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "suite: $(tail -1 $O/pytest_gpu.log)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash profiles/collect.sh r05 > $O/collect.log 2>&1
tail -1 $O/collect.log | cut -c1-400
vr() { timeout 400 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | cut -c1-330; }
vr --virtual-ranks 8 --res 1920x1080 --pictures 129
vr --virtual-ranks 8 --res 3840x2160 --pictures 65
