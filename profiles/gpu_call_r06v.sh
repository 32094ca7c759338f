#!/bin/bash
# GPU call r06v: tile-per-lane SATD, 2 against 4 candidates per lane group (same box, alternating)
tag=${1:-r06v}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for rep in 1 2 3; do for cps in 2 4; do
HAVOC_SATD_TILE_CPS=$cps timeout 400 $B 2>>$O/err.log | tail -1 > $O/b_${cps}_$rep.json; python - <<PY
import json
d=json.load(open("$O/b_${cps}_$rep.json")); print("cps $cps rep $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"]["satd_planes"], d["extra"]["primitives_one_in_flight_latency"]["ms_per_picture"])
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
