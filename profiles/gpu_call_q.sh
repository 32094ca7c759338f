#!/bin/bash
# GPU call Q: k_interp_planes_q with K vertically adjacent tiles per workgroup (HAVOC_PLANES_STRIP = K): parity of the plane
# tests for each K, then the isolated launch time at 1080p.
# (HAVOC_PLANES_STRIP existed only in the experiment this script measured -- slower, not kept; profiles/r02_experiments.md.)
tag=${1:-r02q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for k in 1 2 3 4 8; do
  HAVOC_PLANES_STRIP=$k timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "golden or other_seed or subpel_planes" --timeout 500 -p no:cacheprovider > $O/${tag}_pytest_$k.log 2>&1
  echo "strip $k: $(tail -1 $O/${tag}_pytest_$k.log | cut -c1-200)"
done
for bd in 8 10; do
  for k in 1 2 3 4 6 8; do
    HAVOC_PLANES_STRIP=$k timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --bit-depth $bd --steps 60 --warmup 5 --kernel-reps 50 2> $O/${tag}_${bd}_$k.err | tail -1 > $O/${tag}_${bd}_$k.json
    python - <<PY
import json
r = json.load(open("$O/${tag}_${bd}_$k.json"))
print("bitDepth $bd strip $k:", r["value"], "fps; interp_planes ms", r["whole_step"]["kernel_ms"].get("interp_planes"), "checksum", r.get("checksum") or r["whole_step"].get("checksum"))
PY
  done
done
