#!/bin/bash
# GPU call r05ab: the banded one-sequence pipeline queued by ONE thread (--vr-issue single; contexts without the intra candidates) against a thread per context and
# against whole pictures (with and without the intra candidates): checksums and rates
tag=${1:-r05ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
vr() { timeout 100 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's', d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
vr --virtual-ranks 2 --res 416x240 --pictures 17 --poc-checksums --vr-bands 1
if [ "$2" != "quick" ]; then
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 4
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-no-intra 1
fi
grep -v amdgpu.ids $O/vr.err | tail -4 | cut -c1-400
