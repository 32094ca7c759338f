#!/bin/bash
# GPU call r06c: cycles per wavefront by section of the RDOQ walk kernels (timing build)
tag=${1:-r06c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
HAVOC_MI355X_LIB=$R/profiles/micro/libhavoc_mi355x_timing.so timeout 200 python profiles/micro/rdoq_timing.py > $O/rdoq_timing_1080p.jsonl 2>$O/err.log; cat $O/rdoq_timing_1080p.jsonl
grep -v amdgpu.ids $O/err.log | tail -5 | cut -c1-300
