#!/bin/bash
# GPU call T: the diagonal RDOQ walk (k_rdoq_diag) against the sequential one (HAVOC_RDOQ_DIAG=0): parity tests, the soak, isolated timing.
tag=${1:-r02t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_rdoq.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1; echo "test_rdoq: $(tail -1 $O/${tag}_pytest.log | cut -c1-200)"
grep -E "Error|assert|FAILED" $O/${tag}_pytest.log | head -10
for dg in 8 4 0; do
  HAVOC_RDOQ_DIAG=$dg timeout 200 python profiles/rdoq_bench.py 20 2> $O/${tag}_bench_$dg.err | tail -1 > $O/${tag}_bench_$dg.json
  python -c "
import json; r=json.load(open('$O/${tag}_bench_$dg.json')); print('diag $dg: ms', r['ms'], 'total', r['total_ms'])"
done
timeout 600 python profiles/rdoq_soak.py > $O/${tag}_soak.log 2>&1; echo "soak: $(tail -2 $O/${tag}_soak.log | cut -c1-300)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "reference_library" --timeout 800 -p no:cacheprovider > $O/${tag}_full.log 2>&1; echo "fullsize: $(tail -1 $O/${tag}_full.log | cut -c1-200)"
