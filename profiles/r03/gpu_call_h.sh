#!/bin/bash
# round 3, GPU call H: sub-sample set width against rounds and wall time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
for cfg in "3 1" "4 1" "5 1" "7 1"; do
  set -- $cfg
  for t in 1 4; do
  HAVOC_PICTURE_SUBK=$1 HAVOC_PICTURE_ALT=$2 timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads $t --repeat 5 --expected none > $O/picture_1080p_k$1_a$2_t$t.json 2>/dev/null
  python - <<PY
import json; r=json.load(open('$O/picture_1080p_k$1_a$2_t$t.json'))['picture']; print('1080p K=$1 alt=$2 threads $t:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','rounds_per_step','reruns','satd_jobs')})
PY
  done
done
for k in 3 5; do
HAVOC_PICTURE_SUBK=$k timeout 300 python bench.py --decisions 2 --decision-pictures 8 > $O/dec_1080p_k$k.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p_k$k.json'))['decision_driven_path']; print('1080p decision path K=$k:', r['value'], r.get('pictures_in_flight_8',{}).get('value'), 'alone', r['one_picture_alone_ms'], 'rounds/step', r['rounds_per_step'])
PY
done
