#!/bin/bash
# GPU call r05b: why k_sad4r (94.7 M VALU instructions) takes as long alone as k_sad4w (148.5 M): the calls alone, per form and workgroup size, kernel trace + SQ
# counters of the run form; the GPU halves of the intra-reference-sample pin (k_intra_gather with strong smoothing) and of the intra chain
tag=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_trace_pin.py tests/test_intra_chain.py -m gpu -q -x -p no:cacheprovider > $O/pytest_intra.log 2>&1; echo "intra tests: $(tail -1 $O/pytest_intra.log)"
python profiles/sad4_bench.py calls > $O/sad4_calls.json 2>/dev/null; cat $O/sad4_calls.json
for wv in 4 2 1; do HAVOC_SAD4_RUN_WAVES=$wv python profiles/sad4_bench.py runs > $O/sad4_runs_w$wv.json 2>/dev/null; cat $O/sad4_runs_w$wv.json; done
HAVOC_SAD4_MAX_RUN=32 python profiles/sad4_bench.py runs 2>/dev/null | tee $O/sad4_runs_max32.json
cd /tmp && export TMPDIR=/tmp
D="python $R/profiles/sad4_bench.py runs 4"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $D > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq1 -- $D > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -- $D > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $O/sq3 -- $D > /dev/null 2>&1
cd $R
python - <<PY
import glob, pandas as pd
O = "$O"
f = glob.glob(f"{O}/kt/**/*kernel_stats.csv", recursive=True)
if f:
    st = pd.read_csv(f[0]); print(st[st["Name"].str.contains("k_sad")].to_string())
rows = []
for t in ("sq1", "sq2", "sq3"):
    for g in glob.glob(f"{O}/{t}/**/*counter_collection.csv", recursive=True):
        c = pd.read_csv(g); c = c[c["Kernel_Name"].str.contains("k_sad4r")]
        rows.append(c.groupby("Counter_Name")["Counter_Value"].mean())
if rows:
    s = pd.concat(rows); s.to_csv(f"{O}/sad4r_counters.csv"); print(s.to_string())
PY
rm -rf $O/kt $O/sq1 $O/sq2 $O/sq3
