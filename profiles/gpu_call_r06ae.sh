#!/bin/bash
# GPU call r06ae: the 32x32 forward DCT of 8-bit content on the matrix cores (k_tu_forward_mfma32): parity (goldens, fused == separate, full size), then the step with / without (HAVOC_TU_MFMA=0), same box
tag=${1:-r06ae}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_rdoq.py tests/test_smoke_entry.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-300 | head -8
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
for rep in 1 2 3; do for m in 1 0; do
HAVOC_TU_MFMA=$m timeout 400 $B 2>>$O/err.log | tail -1 > $O/b_${m}_$rep.json; python - <<PY
import json
d=json.load(open("$O/b_${m}_$rep.json")); print("mfma $m rep $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"]["tu_forward"], d["extra"]["primitives_one_in_flight_latency"]["ms_per_picture"])
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
