#!/bin/bash
# round 3, GPU call G: picture client with wider / speculative sub-sample sets and +-72 miss surfaces centred on the first predictor
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_search.py tests/test_decisions.py -m gpu -x -q -k "picture or decision" ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for cfg in "7 1" "3 0" "7 0" "11 1"; do
  set -- $cfg
  HAVOC_PICTURE_SUBK=$1 HAVOC_PICTURE_ALT=$2 timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads 1 --repeat 4 --expected none > $O/picture_1080p_k$1_a$2.json 2>/dev/null
  python - <<PY
import json; r=json.load(open('$O/picture_1080p_k$1_a$2.json'))['picture']; print('1080p K=$1 alt=$2:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','rounds_per_step','reruns','satd_jobs','bytes_down')})
PY
done
for t in 2 4; do
  timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads $t --repeat 4 --expected none > $O/picture_1080p_t$t.json 2>/dev/null
  python - <<PY
import json; r=json.load(open('$O/picture_1080p_t$t.json'))['picture']; print('1080p threads $t:', {k: r[k] for k in ('seconds','rounds','seconds_gpu','seconds_host')})
PY
done
timeout 300 python bench.py --decisions 2 --decision-pictures 8 > $O/dec_1080p.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p.json'))['decision_driven_path']; print('1080p decision path:', r['value'], r.get('pictures_in_flight_8',{}).get('value'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], 'rounds/step', r['rounds_per_step'])
PY
timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 3840x2160 > $O/dec_4k.json 2> $O/dec4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k.json'))['decision_driven_path']; print('4K decision path:', r['value'], r.get('pictures_in_flight_8',{}).get('value'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'], 'rounds/step', r['rounds_per_step'])
PY
