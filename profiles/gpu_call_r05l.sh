#!/bin/bash
# GPU call r05l: k_sad4r with a LANE PER CANDIDATE (U = 0): parity (runs / window tests, source block from LDS and through scalar loads), then the 1080p picture's 1.27 M
# calls timed by form, source path, workgroup size and run caps, and the vector-instruction count of the chosen form
tag=${1:-r05l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for srcv in l g; do
  HAVOC_SAD4_RUN_SRC=$srcv timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest_$srcv.log 2>&1; echo "tests src=$srcv: $(tail -1 $O/pytest_$srcv.log)"; grep -E "^E |^FAILED" $O/pytest_$srcv.log | head -6
done
run() { env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl | cut -c1-230; }
run HAVOC_SAD4_RUN_UNROLL=1
for srcv in l g; do
  for wv in 4 2; do
    for caps in 16,48,128 32,64,128 16,32,64 64,128,128; do
      run HAVOC_SAD4_RUN_SRC=$srcv HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_CAPS=$caps
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for srcv in l g; do
  HAVOC_SAD4_RUN_SRC=$srcv timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_$srcv -- python $R/profiles/sad4_bench.py runs 2 > /dev/null 2>&1
  python - <<PY
import glob, pandas as pd
f = glob.glob("$O/pmc_$srcv/**/*counter_collection.csv", recursive=True)
if f:
    t = pd.read_csv(f[0]); t = t[t["Kernel_Name"].str.contains("k_sad4r")]
    print("src=$srcv", (t.groupby("Counter_Name")["Counter_Value"].sum() / t["Dispatch_Id"].nunique()).round(0).to_dict())
PY
done
