#!/bin/bash
# GPU call r05ac: the frame-parallel GPU tests with the banded pipeline queued both ways, then its 1080p rates
tag=${1:-r05ac}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_frame_parallel.py tests/test_step_banded.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-300 | head -8
vr() { timeout 100 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's', d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-no-intra 1
vr --virtual-ranks 1 --res 1920x1080 --pictures 33 --poc-checksums --vr-bands 4
grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300
