#!/bin/bash
# GPU call r05c: the run form of the 4-way SAD with two calls per pass, and how the calls are cut into runs (work per workgroup); the hooked encoder with the intra stage
# and the TU chains served; the GPU halves of the intra-reference-sample pin and the intra chain (strong smoothing in k_intra_gather)
tag=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_trace_pin.py tests/test_intra_chain.py tests/test_sad4_runs.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "tests a: $(tail -1 $O/pytest_a.log)"
run() { env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl; }
python profiles/sad4_bench.py calls 2>/dev/null | tee -a $O/sad4_variants.jsonl
run HAVOC_SAD4_RUN_UNROLL=1
run HAVOC_SAD4_RUN_UNROLL=2
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_MAX_RUN=64
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_MAX_RUN=32
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_POLICY=64:16,32:32,16:64,8:128
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_POLICY=64:16,32:32,16:128,8:128
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_POLICY=64:32,32:64,16:128,8:128
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_POLICY=64:8,32:16,16:32,8:64
run HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_RUN_WAVES=2 HAVOC_SAD4_POLICY=64:16,32:32,16:64,8:128
timeout 900 python -m pytest tests/test_reference_encoder.py -m gpu -q -x -s -k hooked -p no:cacheprovider > $O/pytest_hooked.log 2>&1; echo "hooked: $(tail -1 $O/pytest_hooked.log)"; grep -E 'us per table call' $O/pytest_hooked.log
