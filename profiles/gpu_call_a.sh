#!/bin/bash
# Round-2 GPU call A: the whole -m gpu suite, the store-mode A/B of k_interp_planes, a default bench line and a serial
# rocprofv3 kernel trace.  Everything lands under gpurun_out/.
tag=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${tag}_pytest.log
tail -30 $O/${tag}_pytest.log
B="python $R/bench.py --no-cpu-baseline --extra-4k 0"
for m in 0 1 2; do
  HAVOC_PLANES_STORE=$m timeout 300 $B --steps 20 --warmup 5 --tune 0 --kernel-reps 20 --min-seconds 0.1 2> $O/${tag}_planes_mode$m.err | tail -1 > $O/${tag}_planes_mode$m.json
  python - <<PY
import json
r=json.load(open("$O/${tag}_planes_mode$m.json"))
print("planes store mode $m: interp_planes ms", r["whole_step"]["kernel_ms"].get("interp_planes"), "GB/s", r["whole_step"]["kernel_gbs"].get("interp_planes"), "step ms", r["ms_per_step"])
PY
done
timeout 600 python bench.py 2> $O/${tag}_bench.err | tail -1 > $O/${tag}_bench.json
cut -c1-1500 $O/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_serial -- $B --steps 10 --warmup 2 --lanes 1 --no-graph --min-seconds 0 > $O/${tag}_serial_bench.log 2>&1
for m in 0 2; do
  HAVOC_PLANES_STORE=$m timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write_mode$m -- $B --steps 2 --warmup 1 --kernel-reps 1 --min-seconds 0 --skip sad4,sad,satd_planes,pred_uni8,satd_inter,pred_uni4,pred_bi8,subtract_bi,pred_bi4,intra_satd35,intra,tu_forward,tu_reconstruct,ssd,recon > /dev/null 2>&1
done
cd $R
find $O/${tag}_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
