#!/bin/bash
# round 3, GPU call R: decision-driven path by pictures in flight and by the number of hardware queues HIP may use (GPU_MAX_HW_QUEUES; default 4)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --decisions 2 --decision-pictures 16 > $O/dec_1080p_q$q.json 2> $O/dec_q$q.err
  python - <<PY
import json; r=json.load(open('$O/dec_1080p_q$q.json'))['decision_driven_path']; print('queues $q 1080p:', r['value'], {k: v['value'] for k, v in r.items() if k.startswith('pictures_in_flight_')}, 'alone', r['one_picture_alone_ms'])
PY
done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --decisions 2 --decision-pictures 16 --res 3840x2160 > $O/dec_4k_q16.json 2> $O/dec4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k_q16.json'))['decision_driven_path']; print('queues 16 4K:', r['value'], {k: v['value'] for k, v in r.items() if k.startswith('pictures_in_flight_')}, 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'])
PY
