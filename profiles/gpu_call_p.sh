#!/bin/bash
# GPU call P: k_interp_planes store/lane-mapping A/B (HAVOC_PLANES_STORE = 2: column per lane + LDS-staged rows, 3: 4 x 4 samples per
# lane), after the whole GPU suite and smoke have passed on the new default.
# (HAVOC_PLANES_STORE was removed from kernels_planes.hip after this measurement, together with the column-per-lane forms it selected.)
tag=${1:-r02p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.log; tail -2 $O/${tag}_smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest.log; tail -4 $O/${tag}_pytest.log | cut -c1-400
for bd in 8 10; do
  for m in 2 3 2 3; do
    HAVOC_PLANES_STORE=$m timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --bit-depth $bd --steps 100 --warmup 5 --kernel-reps 50 2> $O/${tag}_${bd}_$m.err | tail -1 > $O/${tag}_${bd}_$m.json
    python - <<PY
import json
r = json.load(open("$O/${tag}_${bd}_$m.json"))
print("bitDepth $bd mode $m:", r["value"], "fps; interp_planes ms", r["whole_step"]["kernel_ms"].get("interp_planes"), "checksum", r.get("checksum") or r["whole_step"].get("checksum"))
PY
  done
done
