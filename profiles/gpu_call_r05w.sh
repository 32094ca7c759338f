#!/bin/bash
# GPU call r05w: the banded step with each band's launches replayed from a HIP graph: parity again, then the one-sequence rates (whole pictures / bands) at 1080p and 4K
tag=${1:-r05w}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_step_banded.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED|whole step|then B" $O/pytest.log | cut -c1-300 | head -12
vr() { timeout 150 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's busy', d.get('busy_fraction_of_the_contexts'), d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300; }
vr --virtual-ranks 8 --res 416x240 --pictures 17 --poc-checksums --vr-bands 1
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 2
vr --virtual-ranks 8 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4
vr --virtual-ranks 4 --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 2
