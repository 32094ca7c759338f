#!/bin/bash
# GPU call r05o: k_sad4p -- the runs by PERSISTENT workgroups with the next run's box on its way into registers while this run is computed: parity (runs / window tests;
# full-size runs == calls), the 1080p picture's calls timed against a workgroup per run, by workgroups per CU; counters
tag=${1:-r05o}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for v in 1 0; do
  HAVOC_SAD4_RUN_PERSIST=$v timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest_$v.log 2>&1; echo "tests persist=$v: $(tail -1 $O/pytest_$v.log)"; grep -E "^E |^FAILED" $O/pytest_$v.log | head -6
done
run() { echo "$@" | tr '\n' ' '; env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['runs'], d['checksum'])"; }
run HAVOC_SAD4_RUN_PERSIST=0 HAVOC_SAD4_RUN_UNROLL=1
run HAVOC_SAD4_RUN_PERSIST=0
run HAVOC_SAD4_RUN_PERSIST=1
for k in 1 2 3 4 5; do run HAVOC_SAD4_RUN_WGS=$k; done
for caps in 32,64,128 16,64,128 8,32,128; do run HAVOC_SAD4_CAPS=$caps; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc -- python $R/profiles/sad4_bench.py runs 2 > /dev/null 2>&1
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True)
if f:
    t = pd.read_csv(f[0]); t = t[t["Kernel_Name"].str.contains("k_sad4")]
    print("persistent", (t.groupby("Counter_Name")["Counter_Value"].sum() / t["Dispatch_Id"].nunique()).round(0).to_dict())
PY
