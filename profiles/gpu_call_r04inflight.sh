#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04inflight; mkdir -p $O; export TMPDIR=/tmp
for n in 2 3 4; do
  timeout 120 python bench.py --inflight $n --decisions 0 --extra-4k 0 --no-cpu-baseline --traffic 0 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'])"
done
