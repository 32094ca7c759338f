#!/bin/bash
# GPU call Z3: hardware queues x pictures in flight, after the RDOQ kernels got shorter (earlier: more than 4 queues was slower)
tag=${1:-r02z3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for cfg in "4 2" "8 2" "8 3" "6 3" "8 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --inflight $2 --steps 200 --warmup 10 2> $O/${tag}_$1_$2.err | tail -1 > $O/${tag}_$1_$2.json
  python -c "
import json; r=json.load(open('$O/${tag}_$1_$2.json')); print('queues $1 inflight $2:', r['value'], 'fps', r['ms_per_step'], 'ms')"
done
