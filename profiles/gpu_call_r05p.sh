#!/bin/bash
# GPU call r05p: four accumulators per lane in the lane-per-candidate form (the v_sad_u8 chain), workgroup per run and persistent
tag=${1:-r05p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
HAVOC_SAD4_RUN_PERSIST=0 timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "tests: $(tail -1 $O/pytest.log)"; grep -E "^E |^FAILED" $O/pytest.log | head -6
run() { echo "$@" | tr '\n' ' '; env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['runs'], d['checksum'])"; }
run HAVOC_SAD4_RUN_PERSIST=0 HAVOC_SAD4_RUN_UNROLL=1
run HAVOC_SAD4_RUN_PERSIST=0
run HAVOC_SAD4_RUN_PERSIST=0 HAVOC_SAD4_RUN_SRC=l
run HAVOC_SAD4_RUN_PERSIST=1
run HAVOC_SAD4_RUN_PERSIST=0 HAVOC_SAD4_CAPS=32,64,128
