#!/bin/bash
# round 3, GPU call B: the picture client (wavefront order, derived predictors) and the decision-driven step on the GPU: parity tests, then timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_search.py tests/test_decisions.py -m gpu -x -q -k "picture or decision" ) > $O/pytest_picture.log 2>&1
tail -5 $O/pytest_picture.log
for t in 1 4 16; do
  timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads $t --repeat 3 --expected none > $O/picture_1080p_t$t.json 2> $O/picture_1080p_t$t.err
  python - <<PY
import json; r=json.load(open('$O/picture_1080p_t$t.json'))['picture']; print('1080p threads $t:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','bytes_down','rounds_per_step')})
PY
done
timeout 600 python tests/picture_runner.py --device real --res 3840x2160 --threads 16 --repeat 3 --expected none > $O/picture_4k_t16.json 2> $O/picture_4k_t16.err
python - <<PY
import json; r=json.load(open('$O/picture_4k_t16.json'))['picture']; print('4K threads 16:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','bytes_down','rounds_per_step')})
PY
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 6000 $O/bench.json
tail -5 $O/bench.err
