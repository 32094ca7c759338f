#!/bin/bash
# round 4, call q: occupancy variants of the LDS window sad4 (HAVOC_SAD4_WINDOW=1: 1536 B / job; 2: 1024 B, 8 waves / SIMD; 3: 768 B, 8; 4: 1024 B, 7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04q; mkdir -p $O
B="python $R/bench.py --traffic-child 2 --no-graph --inflight 1 --tune 0 --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0"
for mode in 1 2 3 4; do
  export HAVOC_SAD4_WINDOW=$mode
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$mode -- $B > $O/t_$mode.log 2>&1
  f=$(find $O/t_$mode -name "*kernel_stats.csv" | head -1); grep -E "k_sad4w|k_sad<1, 4" $f > $O/stats_$mode.txt
  rm -rf $O/t_$mode
done
