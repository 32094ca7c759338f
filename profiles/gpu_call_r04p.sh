#!/bin/bash
# round 4, call p: the LDS window form of sad4 with dword-aligned reads (odd pitch): parity + duration against the direct kernel, then its LDS counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
( cd $R && timeout 600 python -m pytest tests/test_sad4_window.py -x -q -m gpu 2>&1 | tail -4 ) > $O/pytest.log 2>&1
B="python $R/bench.py --traffic-child 2 --no-graph --inflight 1 --tune 0 --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0"
for mode in win direct; do
  if [ $mode = win ]; then unset HAVOC_SAD4_WINDOW; else export HAVOC_SAD4_WINDOW=0; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$mode -- $B > $O/t_$mode.log 2>&1
  f=$(find $O/t_$mode -name "*kernel_stats.csv" | head -1); grep -E "k_sad4w|k_sad<1, 4" $f > $O/stats_$mode.txt
  rm -rf $O/t_$mode
done
unset HAVOC_SAD4_WINDOW
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/p_1 -- $B > $O/p_1.log 2>&1
python - <<PY > $O/counters.txt 2>&1
import glob, pandas as pd
for g in glob.glob("$O/p_1/**/*counter_collection.csv", recursive=True):
    c=pd.read_csv(g); c=c[c["Kernel_Name"].str.contains("k_sad4w")]
    print(c.groupby("Counter_Name")["Counter_Value"].mean().to_string())
PY
rm -rf $O/p_1
