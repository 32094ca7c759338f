#!/bin/bash
# GPU call r05i: one sequence with its picture dependencies, dataflow between the slots against a barrier, longer sequences (steady state); the bench's parity leg after the
# CPU worker's fix, three times
tag=${1:-r05i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_frame_parallel.py -m gpu -q -x -p no:cacheprovider -k "virtual_ranks" > $O/pytest_a.log 2>&1; echo "tests: $(tail -1 $O/pytest_a.log)"; grep -E "^E " $O/pytest_a.log | head -8
vr() { timeout 400 python bench.py --decisions 4 "$@" 2>>$O/vr.err | tail -1 | tee -a $O/vr.jsonl | cut -c1-330; }
vr --virtual-ranks 8 --res 1920x1080 --pictures 129
vr --virtual-ranks 8 --res 1920x1080 --pictures 129 --vr-dataflow 0
vr --virtual-ranks 4 --res 1920x1080 --pictures 129
vr --virtual-ranks 16 --res 1920x1080 --pictures 129
vr --virtual-ranks 8 --res 3840x2160 --pictures 65
vr --virtual-ranks 8 --res 3840x2160 --pictures 65 --vr-dataflow 0
for i in 1 2 3; do timeout 300 python bench.py --decisions 0 --extra-4k 0 --traffic 0 --detail-out $O/bench_detail_$i.json > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print(d['value'], d['parity'], d['cpu_baseline']['parity_vs_reference'], d['roofline']['kernel'], d['roofline']['launch_ms'])"; done
