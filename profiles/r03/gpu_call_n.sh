#!/bin/bash
# round 3, GPU call N: where a device-resident search spends its time (wall-clock ticks written into the results by -DHAVOC_SEARCH_TIMING builds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
cp turingcodec_amd/libhavoc_mi355x.so /tmp/keep.so
for t in 1 10 12 13; do
  cp turingcodec_amd/libhavoc_timing$t.so turingcodec_amd/libhavoc_mi355x.so
  echo "variant $t"
  timeout 300 python profiles/r03/search_timing_run.py 2>&1 | tail -1 | tee $O/timing_$t.json
done
cp /tmp/keep.so turingcodec_amd/libhavoc_mi355x.so
