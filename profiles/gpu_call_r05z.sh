#!/bin/bash
# GPU call r05z: the banded pipeline's checksums by how the band copies are issued (debug)
tag=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
vr() { timeout 200 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), d.get('between_slots')[:12], d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
for nf in 0 1 2 3; do echo "no_foreach=$nf"; for rep in 1 2; do HAVOC_VR_NO_FOREACH=$nf vr --virtual-ranks 4 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4; done; done
vr --virtual-ranks 4 --res 1920x1080 --pictures 65 --poc-checksums
grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300
