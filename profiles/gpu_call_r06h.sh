#!/bin/bash
# GPU call r06h: the whole -m gpu suite + smoke + the default bench line at the state of round 6's first commits (serve layer for the inter TU chain,
# k_intra_measure, DeviceFrame moved into the package)
tag=${1:-r06h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "gpu suite: $(tail -1 $O/pytest_gpu.log)"; grep -E "^E |^FAILED" $O/pytest_gpu.log | cut -c1-300 | head -8
timeout 600 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc $?"; tail -1 $O/bench.json | cut -c1-1500
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
grep -v amdgpu.ids $O/bench.err | tail -3 | cut -c1-300
