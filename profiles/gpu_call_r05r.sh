#!/bin/bash
# GPU call r05r: the 16-bit path of k_sad4r's lane-per-candidate form (64 samples = 32 dwords per row, unrolled) against the round's first form, 4K Main10 and 1080p 10-bit
tag=${1:-r05r}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | head -6
run() { echo "$@" | tr '\n' ' '; env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['runs'], d['checksum'])"; }
for res in 1920x1080x10 3840x2160x10 3840x2160x8; do
  run HAVOC_SAD4_RES=$res HAVOC_SAD4_RUN_UNROLL=1
  run HAVOC_SAD4_RES=$res
  run HAVOC_SAD4_RES=$res HAVOC_SAD4_RUN_SRC=l
done
