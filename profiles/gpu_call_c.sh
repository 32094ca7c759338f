#!/bin/bash
# Round-2 GPU call C: full -m gpu suite again, search report, SQ counters of the streaming / SATD kernels.
tag=${1:-r02c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${tag}_pytest.log
grep -n "mismatches_by_group\|Error\|passed\|failed" $O/${tag}_pytest.log | cut -c1-600 | head -30
timeout 600 python tests/search_runner.py --device real --res 1920x1080 --searches 5900 --bi 200 --threads 16 > $O/${tag}_search_1080p.json 2> $O/${tag}_search.err
cut -c1-2500 $O/${tag}_search_1080p.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --steps 2 --warmup 1 --kernel-reps 1 --min-seconds 0 --tune 0 --lanes 1 --no-graph"
ONLY="--skip sad4,sad,pred_uni8,satd_inter,pred_uni4,pred_bi8,subtract_bi,pred_bi4,intra_satd35,intra,tu_forward,tu_reconstruct,ssd,recon"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/${tag}_sq1 -- $B $ONLY > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/${tag}_sq2 -- $B $ONLY > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${tag}_tc -- $B $ONLY > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- $B $ONLY > /dev/null 2>&1
cd $R
python - <<PY
import csv,glob,collections
for d in ("sq1","sq2","tc","fetch"):
    for f in glob.glob("$O/${tag}_"+d+"/*/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "havoc_gpu" not in k: continue
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(d, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
