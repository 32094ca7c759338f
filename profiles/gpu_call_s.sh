#!/bin/bash
# GPU call S: blocks per wavefront of the sorted RDOQ walk (HAVOC_RDOQ_PER_WAVE): a wavefront runs as long as its densest block, and
# 162 full wavefronts leave most of the 1024 SIMDs idle -- narrower wavefronts cost nothing and shorten the longest chain.
# (HAVOC_RDOQ_PER_WAVE existed only in the experiment this script measured -- no gain, not kept; profiles/r02_experiments.md.)
tag=${1:-r02s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for pw in 64 32 16 8 4; do
  HAVOC_RDOQ_PER_WAVE=$pw timeout 200 python profiles/rdoq_bench.py 20 2> $O/${tag}_$pw.err | tail -1 > $O/${tag}_$pw.json
  python -c "
import json; r=json.load(open('$O/${tag}_$pw.json')); print('per wave $pw: ms', r['ms'], 'total', r['total_ms'])"
done
HAVOC_RDOQ_PER_WAVE=16 timeout 600 python -m pytest tests/test_rdoq.py -m gpu -q -x --timeout 500 -p no:cacheprovider 2>&1 | tail -1
