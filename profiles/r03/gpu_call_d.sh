#!/bin/bash
# round 3, GPU call D: residual-quadtree batch client + the decision step with it: parity tests, then the decision-driven path
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_search.py tests/test_decisions.py -m gpu -x -q -k "rqt or decision or picture" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cp gpurun_out/rqt_report_1080p.json $O/ 2>/dev/null
python - <<PY
import json; r=json.load(open('$O/rqt_report_1080p.json')); print('1080p rqt:', r['rqt'], r['depth_histogram'], 'expected (1 core) s:', r['expected_seconds'])
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 8 > $O/dec_1080p.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p.json'))['decision_driven_path']; print('1080p decision path:', r['value'], r.get('pictures_in_flight_8'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r['transform_tree_decisions'])
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 3840x2160 > $O/dec_4k.json 2> $O/dec4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k.json'))['decision_driven_path']; print('4K decision path:', r['value'], r.get('pictures_in_flight_8'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r['transform_tree_decisions'])
PY
