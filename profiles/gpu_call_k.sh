#!/bin/bash
# GPU call K: the per-configuration table of DESIGN.md 5 (same code, BASELINE.json configurations), with and without RDOQ in the chain.
tag=${1:-r02k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for cfg in "1920x1080 8 32" "1920x1080 10 32" "3840x2160 8 27" "3840x2160 10 27" "7680x4320 8 32"; do
  set -- $cfg
  for q in 1 0; do
    steps=100; [ "$1" = "7680x4320" ] && steps=20
    timeout 400 python bench.py --no-cpu-baseline --extra-4k 0 --res $1 --bit-depth $2 --qp $3 --rdoq $q --steps $steps --warmup 5 --min-seconds 0.3 2> $O/${tag}_$1_$2_$q.err | tail -1 > $O/${tag}_$1_$2_$q.json
    python - <<PY
import json
try:
    r = json.load(open("$O/${tag}_$1_$2_$q.json"))
    print("$1 $2-bit qp$3 rdoq=$q:", r["value"], "fps", r["ms_per_step"], "ms; rdoq kernel ms", r["whole_step"]["kernel_ms"].get("rdoq"))
except Exception as e:
    print("$1 $2 $q failed", e)
PY
  done
done
timeout 300 python bench.py --mix ai --res 640x360 --qp 32 --extra-4k 0 2> $O/${tag}_ai.err | tail -1 > $O/${tag}_ai.json
python -c "
import json; r=json.load(open('$O/${tag}_ai.json')); print('640x360 all-intra:', r['value'], r['ms_per_step'], (r.get('cpu_baseline') or {}).get('value'), ((r.get('cpu_baseline') or {}).get('parity_vs_reference') or {}).get('mismatches'))"
