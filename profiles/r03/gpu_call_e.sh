#!/bin/bash
# round 3, GPU call E: RDOQ -- 16x16 blocks through the anti-diagonal walk (4 / 8 lanes per block) against the sequential walk: parity, then isolated timings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_rdoq.py -m gpu -x -q ) > $O/pytest_rdoq.log 2>&1
tail -4 $O/pytest_rdoq.log
for f in 0 4 8; do
  HAVOC_RDOQ_DIAG16=$f timeout 300 python profiles/rdoq_bench.py 20 > $O/rdoq_1080p_diag16_$f.json 2>/dev/null
  echo "1080p QP32, 16x16 form $f: $(python -c "import json; r=json.load(open('$O/rdoq_1080p_diag16_$f.json')); print(r['ms'], r['total_ms'], r['groups'])")"
done
for f in 0 4; do
  HAVOC_RDOQ_DIAG16=$f timeout 300 python profiles/rdoq_bench.py 10 3840x2160 27 > $O/rdoq_4k_diag16_$f.json 2>/dev/null
  echo "4K QP27, 16x16 form $f: $(python -c "import json; r=json.load(open('$O/rdoq_4k_diag16_$f.json')); print(r['ms'], r['total_ms'], r['groups'])")"
done
for f in 0 4; do
  HAVOC_RDOQ_DIAG16=$f timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 2> /dev/null | tail -1 > $O/bench_diag16_$f.json
  echo "bench 1080p, 16x16 form $f: $(python -c "import json; r=json.load(open('$O/bench_diag16_$f.json')); print(r['value'], r['ms_per_step'])")"
  HAVOC_RDOQ_DIAG16=$f timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --res 3840x2160 --qp 27 --steps 40 2> /dev/null | tail -1 > $O/bench4k_diag16_$f.json
  echo "bench 4K QP27, 16x16 form $f: $(python -c "import json; r=json.load(open('$O/bench4k_diag16_$f.json')); print(r['value'], r['ms_per_step'])")"
done
