#!/bin/bash
# round 4, call u: 8 / 16 pictures in flight, whole step against searches only (where the in-flight scaling is lost)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp
for ph in all search; do for n in 8 16; do
  HAVOC_DECISION_PHASES=$ph timeout 300 python bench.py --decisions 2 --decision-pictures $n --res 1920x1080 2>/dev/null | tail -1 > $O/dec_${ph}_$n.json
done; done
python - <<PY
import json
for ph in ("all","search"):
  for n in (8,16):
    j=json.loads(open("$O/dec_%s_%d.json"%(ph,n)).read())["decision_driven_path"]
    print(ph, n, j["value"], {k:v["value"] for k,v in j.items() if k.startswith("pictures_in_flight_")})
PY
