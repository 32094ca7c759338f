#!/bin/bash
# GPU call r04e: k_sad4w (window through LDS, row-walking lanes) parity + the step's sad4 time against the direct kernel's unroll / occupancy variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
python -m pytest tests/test_sad4_window.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
B="python bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3"
for v in win direct 2 3 4 5; do
  unset HAVOC_SAD4_DIRECT HAVOC_SAD4_VARIANT
  if [ $v = direct ]; then export HAVOC_SAD4_DIRECT=1; elif [ $v != win ]; then export HAVOC_SAD4_VARIANT=$v; fi
  $B > $O/bench_$v.json 2> $O/err_$v
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$v.json")); print("$v", d["value"], d["ms_per_step"], "sad4 ms", d["whole_step"]["kernel_ms"]["sad4"], d["checksum"])
except Exception as e: print("$v", "ERR", e)
PY
done
