#!/bin/bash
# GPU call R: k_interp_planes_q variant check: plane parity tests, then the isolated launch time at 1080p (8- and 10-bit), twice.
tag=${1:-r02r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_pipeline.py -m gpu -q -x -k "golden or other_seed or subpel_planes or pipeline" --timeout 500 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1
echo "pytest: $(tail -1 $O/${tag}_pytest.log | cut -c1-200)"
for bd in 8 10 8 10; do
  timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --bit-depth $bd --steps 60 --warmup 5 --kernel-reps 50 2> $O/${tag}_${bd}.err | tail -1 > $O/${tag}_${bd}.json
  python - <<PY
import json
r = json.load(open("$O/${tag}_${bd}.json"))
k = r["whole_step"]["kernel_ms"]
print("bitDepth $bd:", r["value"], "fps; interp_planes ms", k.get("interp_planes"), "intra_satd35", k.get("intra_satd35"), "satd_multi", k.get("satd_multi"), "checksum", r.get("checksum") or r["whole_step"].get("checksum"))
PY
done
