#!/bin/bash
# GPU call r05ae: the banded pipeline with its new defaults (24 hardware queues, band streams at the searching streams' priority): the frame-parallel tests, 16 / 20 queues, the rates
tag=${1:-r05ae}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_frame_parallel.py -m gpu -q -p no:cacheprovider -k virtual_ranks > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-300 | head -6
vr() { timeout 60 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's', d.get('between_slots')[-22:], d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
for q in 16 20; do echo "queues $q"; GPU_MAX_HW_QUEUES=$q vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4 --vr-issue threads; done
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 4 --vr-issue threads
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --poc-checksums --vr-bands 4
grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300
