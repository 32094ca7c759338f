#!/bin/bash
# GPU call Z4: ONE picture in flight (the latency of a picture's step, what a dependency-bound encoder sees): diagonal against sequential 32x32 RDOQ walk
tag=${1:-r02z4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2; do
  for dg in 4 0; do
    HAVOC_RDOQ_DIAG=$dg timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --inflight 1 --steps 200 --warmup 10 2> $O/${tag}_${dg}_$rep.err | tail -1 > $O/${tag}_${dg}_$rep.json
    python -c "
import json; r=json.load(open('$O/${tag}_${dg}_$rep.json')); print('one picture in flight, diag $dg run $rep:', r['value'], 'fps', r['ms_per_step'], 'ms')"
  done
done
