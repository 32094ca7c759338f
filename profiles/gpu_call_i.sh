#!/bin/bash
# GPU call I: whole GPU suite + smoke + the default bench line after RDOQ joined the chain.
tag=${1:-r02i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.log; tail -2 $O/${tag}_smoke.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest.log; tail -6 $O/${tag}_pytest.log | cut -c1-400
timeout 600 python bench.py 2> $O/${tag}_bench.err | tail -1 > $O/${tag}_bench.json
python - <<PY
import json
r = json.load(open("$O/${tag}_bench.json"))
print("fps", r["value"], "ms/step", r["ms_per_step"], "roofline", r["roofline"]["kernel"], r["roofline"]["frac"])
print("extra", json.dumps(r.get("extra"))[:900])
c = r.get("cpu_baseline") or {}
print("cpu", c.get("value"), c.get("cores"), (c.get("parity_vs_reference") or {}).get("compared"), (c.get("parity_vs_reference") or {}).get("mismatches"))
PY
