#!/bin/bash
# GPU call r05n: k_sad4r with the ROWS form for 64x64 blocks (a lane per block row, source and box rows in registers, candidates in the order of their box row) beside
# the lane-per-candidate form for the other sizes: parity with each source path, then the 1080p picture's calls timed (rows on / off), counters of each
tag=${1:-r05n}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for v in "1 l" "1 g" "0 g"; do set -- $v
  HAVOC_SAD4_RUN_ROWS=$1 HAVOC_SAD4_RUN_SRC=$2 timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest_$1_$2.log 2>&1; echo "tests rows=$1 src=$2: $(tail -1 $O/pytest_$1_$2.log)"; grep -E "^E |^FAILED" $O/pytest_$1_$2.log | head -6
done
run() { echo "$@" | tr '\n' ' '; env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['runs'], d['checksum'])"; }
run HAVOC_SAD4_RUN_UNROLL=1
for rw in 0 1; do for srcv in l g; do for wv in 4 2; do
  run HAVOC_SAD4_RUN_ROWS=$rw HAVOC_SAD4_RUN_SRC=$srcv HAVOC_SAD4_RUN_WAVES=$wv
done; done; done
for caps in 32,64,128 16,32,128 16,64,128 8,48,128; do run HAVOC_SAD4_RUN_SRC=g HAVOC_SAD4_CAPS=$caps; done
cd /tmp && export TMPDIR=/tmp
for v in "0 g" "1 l" "1 g"; do set -- $v
  HAVOC_SAD4_RUN_ROWS=$1 HAVOC_SAD4_RUN_SRC=$2 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_$1_$2 -- python $R/profiles/sad4_bench.py runs 2 > /dev/null 2>&1
  python - <<PY
import glob, pandas as pd
f = glob.glob("$O/pmc_$1_$2/**/*counter_collection.csv", recursive=True)
if f:
    t = pd.read_csv(f[0]); t = t[t["Kernel_Name"].str.contains("k_sad4r")]
    print("rows=$1 src=$2", (t.groupby("Counter_Name")["Counter_Value"].sum() / t["Dispatch_Id"].nunique()).round(0).to_dict())
PY
done
