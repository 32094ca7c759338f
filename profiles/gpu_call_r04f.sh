#!/bin/bash
# counters of the sad4 kernel chosen by the environment (window form by default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
B="python $R/bench.py --traffic-child 2 --no-graph --inflight 1 --tune 0 --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0"
i=0
for set in "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p_$i -- $B > $O/p_$i.log 2>&1
done
python - <<PY
import glob, pandas as pd
rows=[]
for g in glob.glob("$O/p_*/**/*counter_collection.csv", recursive=True):
    c=pd.read_csv(g); c=c[c["Kernel_Name"].str.contains("k_sad4w|k_sad<1, 4")]
    rows.append(c.groupby(["Kernel_Name","Counter_Name"])["Counter_Value"].mean())
if rows: print(pd.concat(rows).to_string())
PY
rm -rf $O/p_?
