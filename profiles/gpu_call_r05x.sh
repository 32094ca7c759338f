#!/bin/bash
# GPU call r05x: where a picture's time goes in the banded one-sequence pipeline (HAVOC_VR_TRACE): one context alone first
tag=${1:-r05x}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for b in 0 2; do
HAVOC_VR_TRACE=1 timeout 150 python bench.py --decisions 4 --virtual-ranks 1 --res 1920x1080 --pictures 17 --vr-bands $b 2>$O/trace_k1_b$b.err | tail -1 | cut -c1-260
grep "^rank" $O/trace_k1_b$b.err | head -20 | cut -c1-200
done
HAVOC_VR_TRACE=1 timeout 150 python bench.py --decisions 4 --virtual-ranks 2 --res 1920x1080 --pictures 17 --vr-bands 2 2>$O/trace_k2.err | tail -1 | cut -c1-260
grep "^rank" $O/trace_k2.err | sort -k5 -n | head -20 | cut -c1-200
