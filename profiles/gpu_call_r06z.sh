#!/bin/bash
# GPU call r06z: the step's scheduling knobs at the final kernels (same box): pictures in flight x lanes x planner tries
tag=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for cfg in "--inflight 2 --lanes 8 --tune 24" "--inflight 3 --lanes 8 --tune 24" "--inflight 2 --lanes 6 --tune 24" "--inflight 2 --lanes 4 --tune 24" "--inflight 2 --lanes 8 --tune 64" "--inflight 2 --lanes 8 --tune 0" "--inflight 2 --lanes 8 --tune 24"; do
timeout 400 $B $cfg 2>>$O/err.log | tail -1 > $O/b.json; python - <<PY
import json
d=json.load(open("$O/b.json")); print("$cfg step", d["ms_per_step"], d["value"], d["parity"])
PY
done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
