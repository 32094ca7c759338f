#!/bin/bash
# GPU call Y: kernel timeline of the overlapped step (two pictures in flight (the default), graph replay)
tag=${1:-r02y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_trace -- python $R/bench.py --no-cpu-baseline --extra-4k 0 --min-seconds 0 --steps 10 --warmup 3 --kernel-reps 1 > $O/${tag}_bench.log 2>&1
cd $R
f=$(find $O/${tag}_trace -name "*kernel_trace.csv" | head -1)
python profiles/timeline.py $f 170 > $O/${tag}_timeline.txt 2>&1
tail -1 $O/${tag}_bench.log | cut -c1-200
head -190 $O/${tag}_timeline.txt
