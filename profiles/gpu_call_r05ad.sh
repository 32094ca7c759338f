#!/bin/bash
# GPU call r05ad: do the contexts' band streams share hardware queues?  the same pipeline with the side / follower streams at the searching streams' priority, by the number of
# hardware queues HIP may use (GPU_MAX_HW_QUEUES)
tag=${1:-r05ad}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
vr() { timeout 60 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's', d.get('between_slots')[-22:], d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
for q in 24 28 40 48; do echo "queues $q"; HAVOC_VR_SIDE_PRIORITY=0 GPU_MAX_HW_QUEUES=$q vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4 --vr-issue threads; done
grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300
