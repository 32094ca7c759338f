#!/bin/bash
# GPU call r06a: where the overlapped primitive step's 0.70 ms go -- marginal cost of each launch group (--skip), one / two pictures in flight, and the kernel timeline of
# the overlapped step (rocprofv3 --kernel-trace, profiles/timeline.py)
tag=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 0 --traffic 0 --min-seconds 0.3 --steps 100 --warmup 10"
line() { python -c "
import json,sys
l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print('$1', d.get('ms_per_step'), d.get('value'))
except Exception as e: print('$1', 'no line', l[:200])"; }
timeout 200 $B 2>$O/err.log | tee $O/base.json | line base
timeout 200 $B --inflight 1 2>>$O/err.log | line inflight1
timeout 200 $B --inflight 3 2>>$O/err.log | line inflight3
for s in sad4 rdoq "interp_planes,satd_planes" intra_satd35 "pred_bi8,subtract_bi,pred_bi4" "recon,deblock" "tu_forward,rdoq,tu_reconstruct,ssd" intra "pred_uni8,satd_inter,pred_uni4" sad; do
  timeout 200 $B --skip "$s" 2>>$O/err.log | line "skip:$s"
done
timeout 200 $B --lanes 16 2>>$O/err.log | line lanes16
timeout 200 $B --lanes 12 2>>$O/err.log | line lanes12
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- $B --steps 10 --warmup 2 --min-seconds 0 --inflight 1 > $O/trace1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- $B --steps 10 --warmup 2 --min-seconds 0 > $O/trace2.log 2>&1
cd $R
for t in trace1 trace2; do f=$(find $O/$t -name "*kernel_trace.csv" | head -1); python profiles/timeline.py $f 110 > $O/${t}_timeline.txt 2>&1; tail -1 $O/${t}_timeline.txt; done
find $O -name "*.csv" -size +3M -delete
grep -v amdgpu.ids $O/err.log | tail -3 | cut -c1-300
