#!/bin/bash
# GPU call r06n: the step with the 16x16 / 32x32 RDOQ launches in job order or sorted (HAVOC_RDOQ_SORT 0 / 1 / 2 / 3)
tag=${1:-r06n}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for sort in 0 1 2 3; do
HAVOC_RDOQ_SORT=$sort timeout 400 $B 2>>$O/err.log | tail -1 > $O/bench_sort$sort.json; python - <<PY
import json
d=json.load(open("$O/bench_sort$sort.json")); print("sort $sort step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"])
PY
done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
