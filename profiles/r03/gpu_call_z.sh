#!/bin/bash
# round 3, GPU call Z: kernel durations of the device search with the bi-directional refinements (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/profiles/r03/debug_bi.py > $O/trace.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search' in r['Name'] or 'interp' in r['Name']:
            print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
