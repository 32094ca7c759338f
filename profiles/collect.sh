#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r02'
# (the summaries are written on the box into gpurun_out/ and profiles/; copy gpurun_out/r02_*.csv / .json into profiles/)
#
# Pass 1: rocprofv3 --kernel-trace --stats with one lane and no graph, so each kernel's duration is its isolated one
#         (the figure bench.py's roofline.achieved is built from).  Pass 2: the same with the default 8 lanes + graph
#         (durations overlap; kept to show the overlap, not for the roofline).  Passes 3/4: PMC counters, one counter
#         per pass and no trace domains beside them.  Pass 5: the plain bench line with the CPU baseline.
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --min-seconds 0"   # profiled runs: one timed block, no side measurements
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_serial -- $B --steps 10 --warmup 2 --lanes 1 --no-graph > $O/${tag}_serial_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_overlap8 -- $B --steps 10 --warmup 2 > $O/${tag}_overlap8_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- $B --steps 2 --warmup 1 --kernel-reps 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- $B --steps 2 --warmup 1 --kernel-reps 1 > /dev/null 2>&1
# SQ counter passes (two sets that fit the hardware's counter slots); summarised into profiles/<tag>_sq_counters.csv
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${tag}_sq1 -- $B --steps 2 --warmup 1 --kernel-reps 1 --lanes 1 --no-graph > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/${tag}_sq2 -- $B --steps 2 --warmup 1 --kernel-reps 1 --lanes 1 --no-graph > /dev/null 2>&1
cd $R
python profiles/summarize.py --sq $tag $O/${tag}_sq1 $O/${tag}_sq2 > $O/${tag}_sq_summary.txt 2>&1
cp profiles/${tag}_sq_counters.csv $O/ 2>/dev/null
# the traffic table must exist before the final bench line so that roofline.traffic is filled from it
python profiles/summarize.py $tag $O/${tag}_serial $O/${tag}_fetch $O/${tag}_write > $O/${tag}_summary.txt 2>&1
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_hbm_traffic.csv $O/ 2>/dev/null
# the overlapped (8 lanes + graph) kernel statistics, same columns
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/${tag}_overlap8/**/*kernel_stats.csv", recursive=True)
if f:
    st = pd.read_csv(f[0]); st = st[st["Name"].str.contains("havoc_gpu")]
    st["Name"] = st["Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
    st.to_csv("$O/${tag}_kernel_stats_overlap8.csv", index=False)
PY
timeout 120 python profiles/rdoq_bench.py 20 > $O/${tag}_rdoq_isolated.json 2>/dev/null
timeout 400 python bench.py 2> $O/${tag}_bench.err | tail -1 > $O/${tag}_bench.json
tail -1 $O/${tag}_serial_bench.log | cut -c1-400
tail -1 $O/${tag}_overlap8_bench.log | cut -c1-400
cat $O/${tag}_bench.json
