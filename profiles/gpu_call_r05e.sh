#!/bin/bash
# GPU call r05e: runs with the cutter's boxes (staged at once, candidates checked against them) against runs whose box the kernel reduces itself; counters of the boxed form
tag=${1:-r05e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sad4_runs.py tests/test_sad4_window.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "tests a: $(tail -1 $O/pytest_a.log)"; grep -E "^E " $O/pytest_a.log | head -5
run() { env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl; }
python profiles/sad4_bench.py calls 2>/dev/null | tee -a $O/sad4_variants.jsonl
for wv in 2 4; do
  run HAVOC_SAD4_RUN_WAVES=$wv
  run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_BOX=0
  run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_MAX_RUN=128
  run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_MAX_RUN=64
  run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_MAX_RUN=32
  run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_RUN_UNROLL=1
done
cd /tmp && export TMPDIR=/tmp
D="python $R/profiles/sad4_bench.py runs 4"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq1 -- $D > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -- $D > /dev/null 2>&1
cd $R
python - <<PY
import glob, pandas as pd
O = "$O"
rows = []
for t in ("sq1", "sq2"):
    for g in glob.glob(f"{O}/{t}/**/*counter_collection.csv", recursive=True):
        c = pd.read_csv(g); c = c[c["Kernel_Name"].str.contains("k_sad4r")]
        rows.append(c.groupby("Counter_Name")["Counter_Value"].mean())
if rows:
    s = pd.concat(rows); s.to_csv(f"{O}/sad4r_counters.csv"); print(s.to_string())
PY
rm -rf $O/sq1 $O/sq2
