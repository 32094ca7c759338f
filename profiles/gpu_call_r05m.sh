#!/bin/bash
# GPU call r05m: k_sad4r, lane per candidate, lanes in call order or SORTED by box row (fewer LDS bank conflicts?), source block from LDS or through scalar loads: parity of
# each, then the 1080p picture's calls timed against the round's first form, LDS counters of each.  (Its first run measured 8-byte box reads on parity-sorted lanes: 0.49 ms.)
tag=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for v in "c l" "c g" "r l" "r g"; do set -- $v
  HAVOC_SAD4_RUN_ORDER=$1 HAVOC_SAD4_RUN_SRC=$2 timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest_$1_$2.log 2>&1; echo "tests order=$1 src=$2: $(tail -1 $O/pytest_$1_$2.log)"; grep -E "^E |^FAILED" $O/pytest_$1_$2.log | head -6
done
run() { echo "$@" | tr '\n' ' '; env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'], d['runs'], d['checksum'])"; }
run HAVOC_SAD4_RUN_UNROLL=1
for rd in c r; do for srcv in l g; do for wv in 4 2; do
  run HAVOC_SAD4_RUN_ORDER=$rd HAVOC_SAD4_RUN_SRC=$srcv HAVOC_SAD4_RUN_WAVES=$wv
done; done; done
for caps in 32,64,128 16,32,128 24,48,128; do run HAVOC_SAD4_RUN_SRC=g HAVOC_SAD4_RUN_ORDER=r HAVOC_SAD4_CAPS=$caps; done
cd /tmp && export TMPDIR=/tmp
for v in "c g" "r l" "r g"; do set -- $v
  HAVOC_SAD4_RUN_ORDER=$1 HAVOC_SAD4_RUN_SRC=$2 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_$1_$2 -- python $R/profiles/sad4_bench.py runs 2 > /dev/null 2>&1
  python - <<PY
import glob, pandas as pd
f = glob.glob("$O/pmc_$1_$2/**/*counter_collection.csv", recursive=True)
if f:
    t = pd.read_csv(f[0]); t = t[t["Kernel_Name"].str.contains("k_sad4r")]
    print("reads=$1 src=$2", (t.groupby("Counter_Name")["Counter_Value"].sum() / t["Dispatch_Id"].nunique()).round(0).to_dict())
PY
done
