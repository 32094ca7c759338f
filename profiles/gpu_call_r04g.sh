#!/bin/bash
# GPU call r04g: the search kernels with a candidate per lane (patternStep / subpelStep): the trace pin (k_search_list vs the reference encoder's decisions),
# the picture tests (k_search_rows vs the walk over the reference's tables), then one 1080p picture alone at reference distance 1 and 4
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04g}; mkdir -p $O; cd $R
python -m pytest tests/test_trace_pin.py -m gpu -q -x -p no:cacheprovider > $O/pin.log 2>&1; tail -2 $O/pin.log
python -m pytest tests/test_search.py -m gpu -q -x -p no:cacheprovider -k "picture_client_on_the_gpu or other_speed or four_pictures or same_results" > $O/search.log 2>&1; tail -2 $O/search.log
for d in 1 4; do
  python bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 --decision-distance $d > $O/dec_d$d.json 2> $O/dec_d$d.err
  python - <<PY
import json
d=json.load(open("$O/dec_d$d.json"))["decision_driven_path"]
print("distance $d:", d["value"], "pictures/s (4 in flight)", d.get("pictures_in_flight_8",{}).get("value"), "(8)", "alone ms", d["one_picture_alone_ms"], d["one_picture_alone_split_ms"])
PY
done
