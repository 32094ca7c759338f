#!/bin/bash
# round 3, GPU call F: boundary strengths derived on the device (k_derive_bs) + the decision step with its loop filter; then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_deblock.py tests/test_decisions.py -m gpu -x -q ) > $O/pytest_bs.log 2>&1
tail -5 $O/pytest_bs.log
timeout 300 python bench.py --decisions 2 --decision-pictures 8 > $O/dec_1080p.json 2> $O/dec.err
python - <<PY
import json; r=json.load(open('$O/dec_1080p.json'))['decision_driven_path']; print('1080p decision path:', r['value'], r.get('pictures_in_flight_8'), 'alone', r['one_picture_alone_ms'], r['one_picture_alone_split_ms'])
PY
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_reference_encoder.py ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
