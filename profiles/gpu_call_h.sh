#!/bin/bash
# GPU call H: SQ counters of the RDOQ walk kernel alone (profiles/rdoq_bench.py)
tag=${1:-r02h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${tag}_sq1 -- python $R/profiles/rdoq_bench.py 2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/${tag}_sq2 -- python $R/profiles/rdoq_bench.py 2 > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("${tag}_sq1", "${tag}_sq2"):
    fs = glob.glob("$O/%s/*/*counter_collection.csv" % d)
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_rdoq" not in k: continue
        k = k[k.index("k_rdoq"):][:16]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(d, k, {c: round(v / cnt[k][c]) for c, v in acc[k].items()}, "launches", max(cnt[k].values()))
PY
