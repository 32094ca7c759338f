#!/bin/bash
# round 3, GPU call P: the decision-driven path with the searches decided inside the kernel: parity of the step, then pictures/s by pictures in flight
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_decisions.py tests/test_search.py -m gpu -x -q -k "decision or picture" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python bench.py --decisions 2 --decision-pictures 16 > $O/dec_1080p.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p.json'))['decision_driven_path']; print('1080p:', r['value'], {k: v['value'] for k, v in r.items() if k.startswith('pictures_in_flight_')}, 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r.get('searches_by_the_batch_client_alone_ms'))
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 16 --res 3840x2160 > $O/dec_4k.json 2> $O/dec4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k.json'))['decision_driven_path']; print('4K:', r['value'], {k: v['value'] for k, v in r.items() if k.startswith('pictures_in_flight_')}, 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], r.get('searches_by_the_batch_client_alone_ms'))
PY
tail -3 $O/dec.err $O/dec4k.err
