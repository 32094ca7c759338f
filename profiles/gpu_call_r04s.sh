#!/bin/bash
# round 4, call s: search kernels without per-call integer divisions: pin + picture tests, then the decision bench at distance 1 and 4
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_trace_pin.py tests/test_search.py tests/test_decisions.py -q -m gpu 2>&1 | tail -3 > $O/pytest.log
for d in 1 4; do
  timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 --decision-distance $d 2>/dev/null | tail -1 > $O/dec_d$d.json
done
python - <<PY
import json
for d in (1,4):
    j=json.loads(open("$O/dec_d%d.json"%d).read())["decision_driven_path"]
    print(d, j["value"], j["pictures_in_flight_8"]["value"], j["one_picture_alone_ms"], j["one_picture_alone_split_ms"])
PY
