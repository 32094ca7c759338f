#!/bin/bash
# GPU call r05aa: everything the band work touched, at the commit: the decision / search / pipeline / frame-parallel tests, the two new test files, smoke; then the
# one-sequence rates once more (whole pictures / bands; 1, 2, 8 contexts)
tag=${1:-r05aa}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_step_banded.py tests/test_search_gate.py tests/test_frame_parallel.py tests/test_decisions.py tests/test_search.py tests/test_pipeline.py tests/test_smoke_entry.py tests/test_deblock.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | cut -c1-300 | head -12
vr() { timeout 200 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d.get('value'), 'K', d.get('virtual_ranks'), d.get('pictures'), 'pictures', d.get('seconds'), 's busy', d.get('busy_fraction_of_the_contexts'), d.get('between_slots'), d.get('checksum_of_poc_checksums'))
except Exception as e: print('no line', l[:300])"; }
for k in 1 2 8; do
vr --virtual-ranks $k --res 1920x1080 --pictures 65 --poc-checksums
vr --virtual-ranks $k --res 1920x1080 --pictures 65 --poc-checksums --vr-bands 4
done
grep -v amdgpu.ids $O/vr.err | tail -3 | cut -c1-300
