#!/bin/bash
# Round-2 GPU call B: re-run of the full-size parity / frame-parallel tests with per-group mismatch counts, and the
# XCD-pairing / streaming-store A/B of k_interp_planes (time and WRITE_SIZE).
tag=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -p no:cacheprovider -k "reference_library or per_poc or line_contract" > $O/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> $O/${tag}_pytest.log
grep -n "mismatches_by_group\|AssertionError\|passed\|failed" $O/${tag}_pytest.log | cut -c1-1500 | head -20
B="python $R/bench.py --no-cpu-baseline --extra-4k 0"
ONLY="--skip sad4,sad,satd_planes,pred_uni8,satd_inter,pred_uni4,pred_bi8,subtract_bi,pred_bi4,intra_satd35,intra,tu_forward,tu_reconstruct,ssd,recon"
for pair in 0 1 2 3; do
  HAVOC_PLANES_PAIR=$pair timeout 300 $B --steps 20 --warmup 5 --tune 0 --kernel-reps 50 --min-seconds 0.05 $ONLY 2> $O/${tag}_pair$pair.err | tail -1 > $O/${tag}_pair$pair.json
  python - <<PY
import json
r=json.load(open("$O/${tag}_pair$pair.json"))
print("planes pair/nt flags $pair: interp_planes ms (2 launches)", r["whole_step"]["kernel_ms"].get("interp_planes"), "GB/s", r["whole_step"]["kernel_gbs"].get("interp_planes"))
PY
done
cd /tmp && export TMPDIR=/tmp
for pair in 0 1 3; do
  HAVOC_PLANES_PAIR=$pair timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write_pair$pair -- $B --steps 2 --warmup 1 --kernel-reps 1 --min-seconds 0 --tune 0 $ONLY > /dev/null 2>&1
  f=$(find $O/${tag}_write_pair$pair -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
acc=collections.defaultdict(list)
for r in rows:
    if "interp_planes" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE": acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("pair $pair WRITE_SIZE KiB per launch", k, round(sum(v)/len(v),1), "n", len(v))
PY
done
