#!/bin/bash
# GPU call r05v: one sequence with its picture dependencies, the reconstructions entering the DPB mirror band by band (bench.py --decisions 4 --vr-bands R): first small and
# short (checksums against whole pictures), then the rates
tag=${1:-r05v}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
vr() { timeout 120 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), d.get('virtual_ranks'), d.get('pictures'), d.get('seconds'), d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; tail -3 $O/vr.err | cut -c1-300; }
vr --virtual-ranks 2 --res 416x240 --pictures 17 --poc-checksums
vr --virtual-ranks 2 --res 416x240 --pictures 17 --poc-checksums --vr-bands 1
if [ "$2" != "quick" ]; then
vr --virtual-ranks 8 --res 416x240 --pictures 17 --poc-checksums --vr-bands 1
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 2
fi
