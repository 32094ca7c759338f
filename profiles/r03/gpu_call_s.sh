#!/bin/bash
# round 3, GPU call S: wavefronts per workgroup of the device search (4 / 8 / 16): parity and time per picture
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
cp turingcodec_amd/libhavoc_mi355x.so /tmp/keep.so
for w in 4 8 16; do
  cp turingcodec_amd/libhavoc_waves$w.so turingcodec_amd/libhavoc_mi355x.so
  for cfg in "640x360 10" "1920x1080 8"; do
    set -- $cfg
    timeout 300 python tests/picture_runner.py --device real --res $1 --bit-depth $2 --threads 16 > $O/w${w}_$1_$2.json 2> $O/w${w}_$1_$2.err
    python - <<PY
import json
try:
    r = json.load(open('$O/w${w}_$1_$2.json')); d = r['on_device']
    print('waves $w $cfg:', d['seconds'], 'mismatches', d['mismatches'], d['field_equal'], 'steps form', r['on_device_step_launches']['seconds'], r['on_device_step_launches']['mismatches_vs_batch_client'])
except Exception as e:
    print('waves $w $cfg: no report', e); print(open('$O/w${w}_$1_$2.err').read()[-1500:])
PY
  done
done
cp /tmp/keep.so turingcodec_amd/libhavoc_mi355x.so
