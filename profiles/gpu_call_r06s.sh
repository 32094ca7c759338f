#!/bin/bash
# GPU call r06s: the step with 32x32 blocks on the diagonal walk (4 lanes per block, default) against the sequential sorted walk (HAVOC_RDOQ_DIAG=0), same box, alternating
tag=${1:-r06s}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for rep in 1 2 3; do for diag in 4 0 8; do
HAVOC_RDOQ_DIAG=$diag timeout 400 $B 2>>$O/err.log | tail -1 > $O/b_${diag}_$rep.json; python - <<PY
import json
d=json.load(open("$O/b_${diag}_$rep.json")); print("diag $diag rep $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"]["rdoq"], d["extra"]["primitives_one_in_flight_latency"]["ms_per_picture"])
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
