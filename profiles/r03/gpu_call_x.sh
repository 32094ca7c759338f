#!/bin/bash
# round 3, GPU call X: SQ / instruction-cache counters of the search kernels with 16 pictures in flight (is the 90 KB kernel thrashing the 64 KB I-cache?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH"; do
  d=$O/$(echo $set | cut -c1-10 | tr ' ' '_')
  GPU_MAX_HW_QUEUES=16 timeout 250 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/bench.py --decisions 2 --decision-pictures 16 > $d.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('$d/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search_rows' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print({k: (round(v / max(1, n[k])), n[k]) for k, v in acc.items()})
PY
done
