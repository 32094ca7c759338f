#!/bin/bash
# round 4, call t: where a search's time goes now (-DHAVOC_SEARCH_TIMING builds of the device library; profiles/r03/search_timing_run.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04t; mkdir -p $O; export TMPDIR=/tmp
for t in 1 2 3 4 5 12; do
  LIBDEV=libhavoc_timing$t.so timeout 200 python profiles/r03/search_timing_run.py > $O/timing_$t.json 2> $O/timing_$t.err
done
tail -n 2 $O/timing_*.json
