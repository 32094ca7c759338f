#!/bin/bash
# round 3, GPU call V: the device search with announced pattern steps / raster grid, at temporal distances 1 and 4: parity (uni, both forms, bi, stress), time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03v2
mkdir -p $O
cd $R
for cfg in "640x360 10 4 slow" "640x360 8 4 fast" "1920x1080 8 4 medium" "1920x1080 8 1 medium"; do
  set -- $cfg
  timeout 300 python tests/picture_runner.py --device real --res $1 --bit-depth $2 --distance $3 --speed $4 --threads 16 --repeat 2 --stress 6 > $O/p_$1_$2_d$3_$4.json 2> $O/err.txt
  python - <<PY
import json
try:
    r = json.load(open('$O/p_$1_$2_d$3_$4.json')); d = r['on_device']; b = r['on_device_with_bi']
    print('$cfg:', 'batch ok' if r['mismatches'] == 0 else 'BATCH BAD', d['seconds'], 'mismatches', d['mismatches'], d['field_equal'], 'steps', r['on_device_step_launches']['mismatches_vs_batch_client'], 'bi', b['mismatches'], b['uni_mismatches_vs_without_bi'], r['stress'], 'calls/search', round(r['loop_calls'] / r['searches'], 1))
except Exception as e:
    print('$cfg: no report', e); print(open('$O/err.txt').read()[-800:])
PY
done
