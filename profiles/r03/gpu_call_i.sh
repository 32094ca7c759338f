#!/bin/bash
# round 3, GPU call I: RDOQ's scan folded into tu_forward + merged prediction launches: parity, then the step with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "scan_inside or merged_prediction or full_size_results or 4k_qp27" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for rep in 1 2; do
for v in "" "--separate-scan" "--pred classes" "--separate-scan --pred classes"; do
  timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 $v 2> /dev/null | tail -1 > $O/bench.json
  echo "1080p [$v] run $rep: $(python -c "import json; r=json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], r['config']['launches_per_frame'])")"
done
done
for v in "" "--separate-scan --pred classes"; do
  timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --res 3840x2160 --qp 27 --steps 40 $v 2> /dev/null | tail -1 > $O/bench4k.json
  echo "4K QP27 [$v]: $(python -c "import json; r=json.load(open('$O/bench4k.json')); print(r['value'], r['ms_per_step'])")"
done
timeout 120 python profiles/rdoq_bench.py 20 > $O/rdoq_isolated.json 2>/dev/null; cat $O/rdoq_isolated.json | cut -c1-200
