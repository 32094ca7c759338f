#!/bin/bash
# GPU call E: parity of the tile-per-lane satd_multi and its A/B against the lane-row form.
tag=${1:-r02e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_search.py -m gpu -q --timeout 900 -p no:cacheprovider -k "satd or subpel or golden or oracle_other or deterministic or ragged or empty or 640x360" > $O/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${tag}_pytest.log
tail -5 $O/${tag}_pytest.log | cut -c1-400
B="python $R/bench.py --no-cpu-baseline --extra-4k 0"
for t in 0 1; do
  for bd in 8 10; do
  HAVOC_SATD_TILE=$t timeout 300 $B --bit-depth $bd --steps 50 --warmup 5 --tune 8 --kernel-reps 30 --min-seconds 0.2 2> $O/${tag}_tile${t}_$bd.err | tail -1 > $O/${tag}_tile${t}_$bd.json
  python - <<PY
import json
r=json.load(open("$O/${tag}_tile${t}_$bd.json"))
k=r["whole_step"]["kernel_ms"]
print("satd tile form $t bit depth $bd: satd_planes ms", k.get("satd_planes"), "step ms", r["ms_per_step"], "fps", r["value"])
PY
  done
done
