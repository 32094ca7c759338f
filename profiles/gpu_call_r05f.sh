#!/bin/bash
# GPU call r05f: the cutter's run lengths by block size, with the boxes given, by workgroup size
tag=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
run() { env "$@" python profiles/sad4_bench.py runs 2>/dev/null | tee -a $O/sad4_variants.jsonl; }
for wv in 4 2; do
  for caps in 16,48,128 16,64,128 32,64,128 8,32,128 16,32,64 12,48,128 24,64,128 16,48,64 32,128,128; do
    run HAVOC_SAD4_RUN_WAVES=$wv HAVOC_SAD4_CAPS=$caps
  done
done
run HAVOC_SAD4_RUN_WAVES=4 HAVOC_SAD4_RUN_UNROLL=2 HAVOC_SAD4_CAPS=32,64,128
