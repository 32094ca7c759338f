#!/bin/bash
# GPU call Z2: does the diagonal RDOQ walk (shorter chain, more instructions) help the overlapped step?  HAVOC_RDOQ_DIAG=0 against the default, alternating.
tag=${1:-r02z2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for dg in 4 0; do
    HAVOC_RDOQ_DIAG=$dg timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --steps 200 --warmup 10 2> $O/${tag}_${dg}_$rep.err | tail -1 > $O/${tag}_${dg}_$rep.json
    python -c "
import json; r=json.load(open('$O/${tag}_${dg}_$rep.json')); print('diag $dg run $rep:', r['value'], 'fps', r['ms_per_step'], 'ms')"
  done
done
