#!/bin/bash
# round 3, GPU call O: SQ counters of the device-resident search (instructions per wave by class, wait cycles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python $R/profiles/r03/search_timing_run.py > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_FLAT --output-format csv -d $O/sq2 -- python $R/profiles/r03/search_timing_run.py > $O/sq2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ('sq1', 'sq2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob('$O/' + d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_search_step' in r['Kernel_Name']:
                acc[r['Counter_Name']]['sum'] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(d, {k: (round(v['sum']), n[k]) for k, v in acc.items()})
PY
tail -2 $O/sq1.log
