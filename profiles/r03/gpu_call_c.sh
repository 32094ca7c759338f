#!/bin/bash
# round 3, GPU call C: picture client after mapped-memory launches, spinning replay threads, guesses past integer misses
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 300 python tests/picture_runner.py --device real --res 1920x1080 --threads 4 --repeat 2 > $O/picture_1080p_check.json 2> $O/check.err
python - <<PY
import json; r=json.load(open('$O/picture_1080p_check.json')); print('1080p parity: mismatches', r['mismatches'], 'field_equal', r['field_equal'])
PY
for t in 1 2 4 8; do
  timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads $t --repeat 4 --expected none > $O/picture_1080p_t$t.json 2> $O/picture_1080p_t$t.err
  python - <<PY
import json; r=json.load(open('$O/picture_1080p_t$t.json'))['picture']; print('1080p threads $t:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','rounds_per_step')})
PY
done
HAVOC_PICTURE_STAGED=1 timeout 600 python tests/picture_runner.py --device real --res 1920x1080 --threads 1 --repeat 4 --expected none > $O/picture_1080p_staged.json 2>/dev/null
python - <<PY
import json; r=json.load(open('$O/picture_1080p_staged.json'))['picture']; print('1080p staged copies, threads 1:', {k: r[k] for k in ('seconds','rounds','launches','seconds_gpu','seconds_host','rounds_per_step')})
PY
for p in 1 4 8 16; do
  timeout 300 python bench.py --decisions 2 --decision-pictures $p > $O/dec_1080p_p$p.json 2> $O/dec_p$p.err
  python - <<PY
import json; r=json.load(open('$O/dec_1080p_p$p.json'))['decision_driven_path']; print('1080p decision path, $p pictures in flight:', r['value'], 'pictures/s; alone', r['one_picture_alone_ms'], 'ms', r['one_picture_alone_split_ms'], 'threads/picture', r['replay_threads_per_picture'])
PY
done
timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 3840x2160 > $O/dec_4k_p8.json 2> $O/dec_4k.err
python - <<PY
import json; r=json.load(open('$O/dec_4k_p8.json'))['decision_driven_path']; print('4K decision path, 8 pictures in flight:', r['value'], 'pictures/s; alone', r['one_picture_alone_ms'], 'ms', r['one_picture_alone_split_ms'])
PY
