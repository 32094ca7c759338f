#!/bin/bash
# GPU call r04final: the whole -m gpu suite at this commit (log kept), smoke(), the search kernel's kernel-trace + SQ counters at reference distance 1 and 4, then the
# round's profiles and bench line (profiles/collect.sh)
tag=${1:-r04final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "suite: $(tail -1 $O/pytest_gpu.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
for d in 1 4; do
  D="python $R/bench.py --decisions 2 --decision-pictures 1 --res 1920x1080 --decision-distance $d"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/search_d$d -- $D > $O/search_d${d}.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/search_d${d}_sq1 -- $D > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/search_d${d}_sq2 -- $D > /dev/null 2>&1
done
cd $R
python - <<PY
import glob, pandas as pd
O = "$O"
for d in (1, 4):
    f = glob.glob(f"{O}/search_d{d}/**/*kernel_stats.csv", recursive=True)
    if f:
        st = pd.read_csv(f[0]); st = st[st["Name"].str.contains("k_search|k_interp|k_tu|k_rdoq|k_intra|k_pred|k_derive|k_deblock|k_pad|k_level|k_merge|k_satd")]
        st["Name"] = st["Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
        st.to_csv(f"{O}/search_d{d}_kernel_stats.csv", index=False)
        print(st.head(3).to_string())
    rows = []
    for tagc in ("sq1", "sq2"):
        for g in glob.glob(f"{O}/search_d{d}_{tagc}/**/*counter_collection.csv", recursive=True):
            c = pd.read_csv(g)
            c = c[c["Kernel_Name"].str.contains("k_search")]
            c["Kernel_Name"] = c["Kernel_Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
            rows.append(c.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index())
    if rows:
        pd.concat(rows).to_csv(f"{O}/search_d{d}_counters.csv", index=False)
PY
rm -rf $O/search_d*_sq1 $O/search_d*_sq2 $O/search_d1 $O/search_d4
bash profiles/collect.sh r04 > $O/collect.log 2>&1
tail -1 $O/collect.log | cut -c1-300
