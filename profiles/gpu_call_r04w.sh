#!/bin/bash
# round 4, call w: the post-search launch sequences as HIP graphs: decision tests, then 8 / 16 pictures in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04w; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decisions.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log
timeout 300 python bench.py --decisions 2 --decision-pictures 16 --res 1920x1080 2>$O/bench.err | tail -1 > $O/dec_16.json
python - <<PY
import json
j=json.loads(open("$O/dec_16.json").read())["decision_driven_path"]
print(j["value"], {k:v["value"] for k,v in j.items() if k.startswith("pictures_in_flight_")}, j["one_picture_alone_ms"], j["one_picture_alone_split_ms"])
PY
