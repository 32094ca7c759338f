#!/bin/bash
# GPU call r06w: k_sad4r with a run's candidates reduced to the distinct positions: parity (sad4 tests + full size), then the step against a build without it (same box)
tag=${1:-r06w}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-400 | head -8
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for rep in 1 2 3; do for lib in dedup nodedup; do
if [ $lib = nodedup ]; then export HAVOC_MI355X_LIB=$R/profiles/micro/libhavoc_mi355x_nodedup.so; else unset HAVOC_MI355X_LIB; fi
timeout 400 $B 2>>$O/err.log | tail -1 > $O/b_${lib}_$rep.json; python - <<PY
import json
d=json.load(open("$O/b_${lib}_$rep.json")); print("$lib rep $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"]["sad4"], d["roofline"]["valu"], d["extra"]["primitives_one_in_flight_latency"]["ms_per_picture"])
PY
done; done
unset HAVOC_MI355X_LIB
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
