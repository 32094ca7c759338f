#!/bin/bash
# GPU call r06l: RDOQ with scan-order masks + counted zero groups: parity, isolated timing; kernel timeline of the five rdoq launches side by side
tag=${1:-r06l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_rdoq.py -m gpu -q -x -p no:cacheprovider > $O/pytest_rdoq.log 2>&1; echo "test_rdoq: $(tail -1 $O/pytest_rdoq.log)"; grep -E "^E |^FAILED" $O/pytest_rdoq.log | cut -c1-300 | head -6
timeout 120 python profiles/rdoq_bench.py 20 > $O/rdoq_isolated.json 2>$O/rdoq_isolated.err; cat $O/rdoq_isolated.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/profiles/micro/rdoq_overlap.py > $O/overlap.json 2>$O/trace.log; cat $O/overlap.json
f=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); echo "trace: $f"
python $R/profiles/timeline.py $f 40 > $O/timeline.txt 2>&1; head -60 $O/timeline.txt
rm -rf $O/trace
