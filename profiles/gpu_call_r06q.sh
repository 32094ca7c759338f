#!/bin/bash
# GPU call r06q: the step under GPU_MAX_HW_QUEUES 2 / 4 (default) / 8 / 16 and 1 / 2 / 4 pictures in flight (same box)
tag=${1:-r06q}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for q in 4 2 8 16; do for inf in 2 1 4; do
GPU_MAX_HW_QUEUES=$q timeout 400 $B --inflight $inf 2>>$O/err.log | tail -1 > $O/b.json; python - <<PY
import json
d=json.load(open("$O/b.json")); print("queues $q inflight $inf step", d["ms_per_step"], d["value"], d["parity"])
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
