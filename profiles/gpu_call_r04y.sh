#!/bin/bash
# round 4, call y: the frame-parallel pipeline driving the decision step: 1 rank (RCCL exchange, whole pictures and bands) against a 2-rank rehearsal on the one GPU (gloo)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04y; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --decisions 3 --gpus 1 --exchange --res 416x240 --pictures 17 2>$O/w1.err | grep "^{" | tail -1 > $O/w1.json
timeout 300 python bench.py --decisions 3 --gpus 1 --exchange --bands 2 --res 416x240 --pictures 17 2>$O/w1b.err | grep "^{" | tail -1 > $O/w1b.json
HAVOC_BENCH_BACKEND=gloo timeout 600 python bench.py --decisions 3 --gpus 2 --res 416x240 --pictures 17 2>$O/w2.err | grep "^{" | tail -1 > $O/w2.json
HAVOC_BENCH_BACKEND=gloo timeout 600 python bench.py --decisions 3 --gpus 2 --bands 2 --res 416x240 --pictures 17 2>$O/w2b.err | grep "^{" | tail -1 > $O/w2b.json
timeout 300 python bench.py --decisions 3 --gpus 1 --exchange --bands 4 --res 1920x1080 --pictures 17 2>$O/w1_1080.err | grep "^{" | tail -1 > $O/w1_1080.json
python - <<PY
import json
r={}
for k in ("w1","w1b","w2","w2b","w1_1080"):
    try:
        j=json.loads(open("$O/%s.json"%k).read()); r[k]=j
        print(k, j["value"], j["pictures"], j["slots"], j["seconds"], j["config"]["exchange"], j["config"]["broadcasts"], j["checksum_of_poc_checksums"])
    except Exception as e:
        print(k, "FAILED", e); print(open("$O/%s.err"%k).read()[-1500:])
if all(k in r for k in ("w1","w1b","w2","w2b")):
    print("per-POC checksums equal:", r["w1"]["poc_checksums"]==r["w1b"]["poc_checksums"]==r["w2"]["poc_checksums"]==r["w2b"]["poc_checksums"], len(r["w1"]["poc_checksums"]))
PY
