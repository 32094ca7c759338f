cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python $R/bench.py --traffic-child 2 --no-graph --inflight 1 --tune 0 --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0"
for mode in win direct; do
  if [ $mode = direct ]; then export HAVOC_SAD4_DIRECT=1; else unset HAVOC_SAD4_DIRECT; fi
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/${mode}_$i -- $B > $O/${mode}_$i.log 2>&1
  done
done
python - <<PY
import glob, pandas as pd
for mode in ("win","direct"):
    rows=[]
    for g in glob.glob(f"$O/{mode}_*/**/*counter_collection.csv", recursive=True):
        c=pd.read_csv(g); c=c[c["Kernel_Name"].str.contains("k_sad4w|k_sad<1, 4>|k_sadILi1ELi4")]
        rows.append(c.groupby("Counter_Name")["Counter_Value"].mean())
    if rows: print(mode); print(pd.concat(rows).to_string())
PY
rm -rf $O/win_? $O/direct_?/*/*agent* 2>/dev/null
