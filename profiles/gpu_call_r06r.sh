#!/bin/bash
# GPU call r06r: the whole -m gpu suite + the default bench line
tag=${1:-r06r}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; echo "gpu suite: $(tail -1 $O/pytest_gpu.log)"; grep -E "^E |^FAILED|^gpu_ra|^ra_medium" $O/pytest_gpu.log | cut -c1-400 | head -12
timeout 600 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc $?"; tail -1 $O/bench.json | cut -c1-600
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
grep -v amdgpu.ids $O/bench.err | tail -3 | cut -c1-300
