#!/bin/bash
# GPU call r05h: one sequence with its picture dependencies on K virtual ranks (bench.py --decisions 4): the test, then 1080p / 4K rates for K = 1, 4, 8; the search test whose
# accounting changed; the bench line with the dominant kernel chosen per launch
tag=${1:-r05h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_frame_parallel.py tests/test_search.py -m gpu -q -x -p no:cacheprovider -k "virtual_ranks or decisions_on_the_gpu" > $O/pytest_a.log 2>&1; echo "tests: $(tail -1 $O/pytest_a.log)"; grep -E "^E " $O/pytest_a.log | head -8
for k in 1 4 8; do timeout 300 python bench.py --decisions 4 --virtual-ranks $k --res 1920x1080 --pictures 33 2>$O/vr_1080p_$k.err | tail -1 | tee $O/vr_1080p_$k.json | cut -c1-400; done
for k in 8; do timeout 400 python bench.py --decisions 4 --virtual-ranks $k --res 3840x2160 --pictures 17 2>$O/vr_4k_$k.err | tail -1 | tee $O/vr_4k_$k.json | cut -c1-400; done
timeout 400 python bench.py --decisions 0 --extra-4k 0 --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; cut -c1-1200 $O/bench.json
