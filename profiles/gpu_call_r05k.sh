#!/bin/bash
# GPU call r05k: where the device-resident walk leaves the host walk with the five-candidate derivation (debug)
tag=${1:-r05k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for res in 416x240 640x360; do
python tests/picture_runner.py --device real --res $res --bit-depth 8 --threads 16 > $O/pic_$res.json 2> $O/pic_$res.err
python - <<PY
import json
r=json.loads(open("$O/pic_$res.json").read().strip().splitlines()[-1])
d=r["on_device"]; print("$res", {k:d[k] for k in d if k not in ("mismatching_searches",)}, r.get("on_device_step_launches"), "batch vs walk:", r.get("mismatches"))
print(d.get("mismatching_searches"))
PY
done
