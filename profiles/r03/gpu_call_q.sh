#!/bin/bash
# round 3, GPU call Q: which SQ counters exist (instruction fetch, wait reasons), and their values for the search kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|SQC_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > $O/avail.txt
wc -w $O/avail.txt
for set in "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM"; do
  d=$O/$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/profiles/r03/search_timing_run.py > $d.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob('$d/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_search_rows' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
print({k: round(v / 3 / 4080) for k, v in acc.items()} or open('$d.log').read()[-300:])
PY
done
