#!/bin/bash
# GPU call r06i: the hooked reference encoder on the MI355X with the report's wait time / waits and launches by entry point; the replay stress test (fixed import)
tag=${1:-r06i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_reference_encoder.py -m gpu -q -x -s -p no:cacheprovider -k "hooked" > $O/pytest_hooked.log 2>&1; echo "hooked: $(tail -1 $O/pytest_hooked.log)"; grep -E "^gpu_ra|^E |^FAILED" $O/pytest_hooked.log | cut -c1-400 | head -8
python - <<'PY' > $O/report.txt 2>&1
import os, sys, tempfile, time
sys.path.insert(0, 'tests')
import encoder_tools as et
for case in ("gpu_ra_medium_qp32", "gpu_ra_medium_10bit"):
    wd = tempfile.mkdtemp()
    t0 = time.perf_counter()
    got, err = et.encode(et.HOOKED_EXE, case, wd, [], env={"HAVOC_CLASSIC_REPORT": "1"}, tag=".hk", timeout=900)
    print(case, f"{time.perf_counter() - t0:.2f} s")
    print('\n'.join(l for l in err.splitlines() if 'libhavoc_classic' in l))
PY
cat $O/report.txt | cut -c1-600
timeout 900 python -m pytest tests/test_replay_stress.py tests/test_intra_measure.py tests/test_smoke_entry.py -m gpu -q -x -p no:cacheprovider > $O/pytest_rest.log 2>&1; echo "rest: $(tail -1 $O/pytest_rest.log)"; grep -E "^E |^FAILED" $O/pytest_rest.log | cut -c1-300 | head -8
