#!/bin/bash
# diagnostic: the row kernel with a lag of one CTU between rows (what the restated predictor derivation's data needs) against the reference's two: same decisions, fewer steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04lag; mkdir -p $O; export TMPDIR=/tmp
for lag in 2 1; do
  HAVOC_SEARCH_ROW_LAG=$lag timeout 200 python tests/picture_runner.py --device real --res 1920x1080 --repeat 3 > $O/picture_lag$lag.json 2> $O/err$lag.txt
  HAVOC_SEARCH_ROW_LAG=$lag timeout 200 python bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 2>/dev/null | grep "^{" | tail -1 > $O/dec_lag$lag.json
done
python - <<PY
import json
for lag in (2,1):
    r=json.load(open("$O/picture_lag%d.json"%lag)); d=json.load(open("$O/dec_lag%d.json"%lag))["decision_driven_path"]
    print("lag", lag, "mismatches vs the walk through the reference's tables", r["on_device"].get("mismatches"), "field equal", r["on_device"].get("field_equal"), "searches", r.get("searches"),
          "| alone ms", d["one_picture_alone_ms"], "search ms", d["one_picture_alone_split_ms"]["searches_in_wavefront_order"], "4 / 8 in flight", d["value"], d["pictures_in_flight_8"]["value"])
PY
