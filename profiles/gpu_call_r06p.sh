#!/bin/bash
# GPU call r06p: same-box A/B of the step: round 5's RDOQ kernels (profiles/micro/libhavoc_mi355x_base.so: kernels_rdoq.hip of commit 505cac7, every other object current)
# against the current ones, alternating, three times each
tag=${1:-r06p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for rep in 1 2 3; do
for lib in base current; do
if [ $lib = base ]; then export HAVOC_MI355X_LIB=$R/profiles/micro/libhavoc_mi355x_base.so; else unset HAVOC_MI355X_LIB; fi
timeout 400 $B 2>>$O/err.log | tail -1 > $O/bench_${lib}_$rep.json; python - <<PY
import json
d=json.load(open("$O/bench_${lib}_$rep.json")); print("$lib $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"]["rdoq"], d["extra"]["primitives_one_in_flight_latency"]["ms_per_picture"])
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
