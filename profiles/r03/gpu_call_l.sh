#!/bin/bash
# round 3, GPU call L: the same with the intra decisions taken on the device (kernels_decide.hip): parity, then timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests/test_search.py tests/test_decisions.py -m gpu -x -q -k "intra_rd or decision" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
cp gpurun_out/intra_rd_report_1080p.json $O/ 2>/dev/null
python - <<PY
import json; r=json.load(open('$O/intra_rd_report_1080p.json')); print({k: (v['partitions'], v['candidates'], v['seconds_batch'], v['seconds_per_call_one_core'], v['champion_is_first_candidate']) for k, v in r['sizes'].items()}, r['device_decisions'])
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 8 > $O/dec_1080p.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p.json'))['decision_driven_path']; print('1080p decision path:', r['value'], r.get('pictures_in_flight_8',{}).get('value'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r['intra'])
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 3840x2160 > $O/dec_4k.json 2> $O/dec4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k.json'))['decision_driven_path']; print('4K decision path:', r['value'], r.get('pictures_in_flight_8',{}).get('value'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r['intra'])
PY
