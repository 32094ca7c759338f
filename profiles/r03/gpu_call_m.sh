#!/bin/bash
# round 3, GPU call M: the motion searches of a picture with the decision loops inside the kernel (kernels_search.hip): parity with the walk over the
# reference's tables, time per picture
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
for cfg in "640x360 8" "640x360 10" "1920x1080 8"; do
  set -- $cfg
  timeout 600 python tests/picture_runner.py --device real --res $1 --bit-depth $2 --threads 16 > $O/picture_$1_$2.json 2> $O/picture_$1_$2.err
  echo "rc=$? $cfg"
  python - <<PY
import json
try:
    r = json.load(open('$O/picture_$1_$2.json'))
    print({k: r.get(k) for k in ('searches', 'mismatches', 'field_equal', 'expected_seconds')}, 'batch client', r['picture']['seconds'], r['on_device'], r['on_device_step_launches'], r.get('on_device_with_bi'))
except Exception as e:
    print('no report', e); print(open('$O/picture_$1_$2.err').read()[-2000:])
PY
done
