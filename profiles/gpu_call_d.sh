#!/bin/bash
# Round-2 GPU call D: parity of the changed kernels, then A/B of satd_multi's tiles-per-lane and the new interp_planes mapping.
tag=${1:-r02d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -m gpu -q --timeout 900 -p no:cacheprovider -k "planes or satd or subpel or golden or oracle_other or deterministic or ragged" > $O/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${tag}_pytest.log
tail -5 $O/${tag}_pytest.log | cut -c1-400
B="python $R/bench.py --no-cpu-baseline --extra-4k 0"
for tpl in 1 2; do
  HAVOC_SATD_TPL=$tpl timeout 300 $B --steps 50 --warmup 5 --tune 8 --kernel-reps 30 --min-seconds 0.2 2> $O/${tag}_tpl$tpl.err | tail -1 > $O/${tag}_tpl$tpl.json
  python - <<PY
import json
r=json.load(open("$O/${tag}_tpl$tpl.json"))
k=r["whole_step"]["kernel_ms"]
print("satd TPL $tpl: satd_planes ms", k.get("satd_planes"), "interp_planes ms", k.get("interp_planes"), "step ms", r["ms_per_step"], "fps", r["value"])
PY
done
