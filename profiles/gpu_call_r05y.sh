#!/bin/bash
# GPU call r05y: one sequence with its picture dependencies on virtual ranks -- whole pictures against bands (bench.py --decisions 4 --vr-bands R): checksums and rates
tag=${1:-r05y}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
vr() { timeout 200 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's busy', d.get('busy_fraction_of_the_contexts'), d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; grep -v amdgpu.ids $O/vr.err | tail -2 | cut -c1-300; }
vr --virtual-ranks 4 --res 1920x1080 --pictures 129 --poc-checksums
vr --virtual-ranks 4 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 4
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 4
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 6
vr --virtual-ranks 2 --res 1920x1080 --pictures 65 --poc-checksums
vr --virtual-ranks 2 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4
