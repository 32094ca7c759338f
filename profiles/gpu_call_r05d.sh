#!/bin/bash
# GPU call r05d: how a search's calls are best cut into runs (work per workgroup evened out by block size), by workgroup size
tag=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
run() { env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl; }
for wv in 4 2; do
  for pol in 64:32,32:64,16:128,8:128 64:16,32:64,16:128,8:128 64:32,32:128,16:128,8:128 64:24,32:48,16:96,8:128 64:16,32:32,16:128,8:128 64:32,32:64,16:64,8:128 64:16,32:48,16:128,8:128 64:12,32:32,16:64,8:128 64:8,32:32,16:64,8:128 64:16,32:16,16:64,8:128; do
    run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_POLICY=$pol
  done
done
