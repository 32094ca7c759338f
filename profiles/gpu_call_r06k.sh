#!/bin/bash
# GPU call r06k: RDOQ parity + isolated timing (per transform size) + the step
tag=${1:-r06k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_rdoq.py -m gpu -q -x -p no:cacheprovider > $O/pytest_rdoq.log 2>&1; echo "test_rdoq: $(tail -1 $O/pytest_rdoq.log)"; grep -E "^E |^FAILED" $O/pytest_rdoq.log | cut -c1-300 | head -6
timeout 120 python profiles/rdoq_bench.py 20 > $O/rdoq_isolated.json 2>$O/rdoq_isolated.err; cat $O/rdoq_isolated.json | cut -c1-200
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
timeout 400 $B 2>>$O/err.log | tail -1 > $O/bench.json; python - <<PY
import json
d=json.load(open("$O/bench.json")); print("step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"])
PY
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
