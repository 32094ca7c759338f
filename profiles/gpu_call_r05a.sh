#!/bin/bash
# GPU call r05a: round 5's fixes first -- smoke() inside the suite, the run form of the 4-way SAD against the oracle, the generic-width guard of k_sad4w,
# the replayed step under stress (VERDICT r4 next #1a), then the bench line twice (runs / per-call form of sad4) with its full-size parity check
tag=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_smoke_entry.py tests/test_sad4_runs.py tests/test_sad4_window.py tests/test_replay_stress.py -m gpu -q -x -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "new tests: $(tail -1 $O/pytest_new.log)"; grep -E 'FAILED|Error|assert' $O/pytest_new.log | head -20
for form in runs calls; do
  timeout 600 python bench.py --steps 20 --warmup 5 --sad4 $form --detail-out $O/bench_detail_$form.json --parity-dump $O/parity_$form.json > $O/bench_$form.json 2> $O/bench_$form.err
  echo "bench $form rc=$? bytes=$(wc -c < $O/bench_$form.json)"; cut -c1-1500 $O/bench_$form.json; tail -3 $O/bench_$form.err
done
