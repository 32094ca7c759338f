#!/bin/bash
# GPU call r06e: RDOQ walks with a smaller LDS footprint (no per-position prefix array, context states read from the CTU snapshot, prefix tables sized by the block,
# scan arrays sharing the walk's memory), straight-line firstInGroup, verdict's clear loop over coded groups only, k_rdoq_diag at two wavefronts per SIMD
tag=${1:-r06e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_rdoq.py -m gpu -q -x -p no:cacheprovider > $O/pytest_rdoq.log 2>&1; echo "test_rdoq: $(tail -1 $O/pytest_rdoq.log)"; grep -E "^E |^FAILED" $O/pytest_rdoq.log | cut -c1-300 | head -6
timeout 120 python profiles/rdoq_bench.py 20 > $O/rdoq_isolated.json 2>$O/rdoq_isolated.err; cat $O/rdoq_isolated.json | cut -c1-200
HAVOC_RDOQ_LDS_STATES=1 timeout 120 python profiles/rdoq_bench.py 20 > $O/rdoq_isolated_lds_states.json 2>$O/rdoq_isolated.err; cat $O/rdoq_isolated_lds_states.json | cut -c1-200
HAVOC_MI355X_LIB=$R/profiles/micro/libhavoc_mi355x_timing.so timeout 200 python profiles/micro/rdoq_timing.py > $O/rdoq_timing_1080p.jsonl 2>$O/err.log; cat $O/rdoq_timing_1080p.jsonl
timeout 200 python profiles/micro/tu_chain_concurrency.py > $O/conc.json 2>>$O/err.log; cat $O/conc.json
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
timeout 400 $B 2>>$O/err.log | tail -1 > $O/bench.json; python - <<PY
import json
d=json.load(open("$O/bench.json")); print("step", d["ms_per_step"], d["value"], d["whole_step"]["kernel_ms"])
PY
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
