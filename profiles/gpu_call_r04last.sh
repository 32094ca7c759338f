#!/bin/bash
# round 4, last call: the plain bench line (with the in-run HBM traffic of the dominant kernel) and the search kernels' statistics at the final commit
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04last; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 1 4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/search_d$d -- python $R/bench.py --decisions 2 --decision-pictures 1 --res 1920x1080 --decision-distance $d > $O/search_d${d}.log 2>&1
done
cd $R
python - <<PY
import glob, pandas as pd
O = "$O"
for d in (1, 4):
    f = glob.glob(f"{O}/search_d{d}/**/*kernel_stats.csv", recursive=True)
    if f:
        st = pd.read_csv(f[0]); st = st[st["Name"].str.contains("k_search|k_interp|k_tu|k_rdoq|k_intra|k_pred|k_derive|k_deblock|k_pad|k_level|k_merge|k_satd|k_rqt|k_block")]
        st["Name"] = st["Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
        st.to_csv(f"{O}/search_d{d}_kernel_stats.csv", index=False)
        print(st.head(3).to_string())
PY
rm -rf $O/search_d1 $O/search_d4
timeout 900 python bench.py 2> $O/bench.err | grep "^{" | tail -1 > $O/r04_bench.json
python - <<PY
import json
d=json.load(open("$O/r04_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["frac"], d["roofline"]["valu_busy_pct"])
for k,v in d["extra"].items():
    if isinstance(v,dict) and "sop_weighted" in v: print(k[:50], v["sop_weighted"]["value"], v["sop_weighted"]["rates_by_reference_distance"], v["one_picture_alone_ms"], v["parity_vs_reference"]["mismatching"], v["parity_vs_reference"]["bi_directional_mismatching"])
PY
