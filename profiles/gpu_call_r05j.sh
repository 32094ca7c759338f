#!/bin/bash
# GPU call r05j: the decision path with the reference's five-candidate predictor derivation (amvp.hpp) in the walk: every test that compares the device search with the
# CPU walk, smoke, and the decision path's rates at 1080p (d1, d4)
tag=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_search.py tests/test_decisions.py tests/test_pipeline.py tests/test_smoke_entry.py tests/test_frame_parallel.py tests/test_trace_pin.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "tests: $(tail -1 $O/pytest_a.log)"; grep -E "^E |^FAILED" $O/pytest_a.log | head -8
for d in 1 4; do timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 --decision-distance $d --decision-walk 1 2>$O/dec_d$d.err | tail -1 > $O/dec_d$d.json; python -c "
import json; d=json.load(open('$O/dec_d$d.json')); p=d['decision_driven_path']; print('d$d', p['one_picture_alone_ms'], p['value'], p.get('pictures_in_flight_8'), d.get('decision_walk',{}).get('parity_vs_reference'))"; done
