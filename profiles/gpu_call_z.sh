#!/bin/bash
# GPU call Z: pictures in flight x lanes of the overlapped step after the RDOQ changes
tag=${1:-r02zz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for cfg in "2 8" "3 8" "4 8" "3 4" "2 12"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --inflight $1 --lanes $2 --steps 200 --warmup 10 2> $O/${tag}_$1_$2.err | tail -1 > $O/${tag}_$1_$2.json
  python -c "
import json; r=json.load(open('$O/${tag}_$1_$2.json')); print('inflight $1 lanes $2:', r['value'], 'fps', r['ms_per_step'], 'ms')"
done
