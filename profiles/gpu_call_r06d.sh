#!/bin/bash
# GPU call r06d: the TU chain's launch groups with the five transform sizes side by side (4 and 8 hardware queues)
tag=${1:-r06d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 200 python profiles/micro/tu_chain_concurrency.py > $O/conc.json 2>$O/err.log; cat $O/conc.json
GPU_MAX_HW_QUEUES=8 timeout 200 python profiles/micro/tu_chain_concurrency.py > $O/conc_q8.json 2>>$O/err.log; cat $O/conc_q8.json
grep -v amdgpu.ids $O/err.log | tail -5 | cut -c1-300
