#!/bin/bash
# GPU call r06y: tile-per-lane SATD for 9 / 10-bit samples: parity (goldens incl. the extreme-difference planes, full size 4K Main10), 4K Main10 step with / without
tag=${1:-r06y}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_smoke_entry.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-300 | head -8
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 50 --warmup 5 --res 3840x2160 --bit-depth 10 --qp 27"
for rep in 1 2; do for tile in 1 0; do
HAVOC_SATD_TILE=$tile timeout 600 $B 2>>$O/err.log | tail -1 > $O/b_${tile}_$rep.json; python - <<PY
import json
d=json.load(open("$O/b_${tile}_$rep.json")); print("4K Main10 tile $tile rep $rep step", d["ms_per_step"], d["value"], d["parity"], d["whole_step"]["kernel_ms"].get("satd_planes"))
PY
done; done
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
