#!/bin/bash
# round 3, GPU call T: smoke, the whole -m gpu suite, the plain bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
r = json.load(open('$O/bench.json'))
print({k: r[k] for k in ('metric', 'value', 'ms_per_step')}, r['roofline'])
for k, v in r.get('extra', {}).items():
    if k.startswith('decision'):
        print(k, v.get('value'), {a: b['value'] for a, b in v.items() if a.startswith('pictures_in_flight_')}, v.get('one_picture_alone_ms'), v.get('error'))
print(r['cpu_baseline'].get('decision_walk'))
PY
