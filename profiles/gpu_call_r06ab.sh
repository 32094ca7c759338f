#!/bin/bash
# GPU call r06ab: kernel time and vector instructions of the DECISION path with 8 pictures in flight (distance 1 and 4): where its instructions go
tag=${1:-r06ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 1 4; do
B="python $R/bench.py --no-cpu-baseline --extra-4k 0 --decisions 2 --decision-distance $d --decision-pictures 8 --traffic 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_d$d -- $B > $O/bench_d$d.log 2>&1; tail -1 $O/bench_d$d.log | cut -c1-300
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_d$d -- $B > /dev/null 2>&1
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/stats_d$d/**/*kernel_stats.csv", recursive=True)
st = pd.read_csv(f[0]); st = st[st["Name"].str.contains("havoc_gpu")].copy()
st["Name"] = st["Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
st = st.sort_values("TotalDurationNs", ascending=False)
st.to_csv("$O/kernel_stats_d$d.csv", index=False)
print(st[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].head(14).to_string())
c = glob.glob("$O/pmc_d$d/**/*counter_collection.csv", recursive=True)
if c:
    df = pd.read_csv(c[0]); df = df[df["Kernel_Name"].str.contains("havoc_gpu")]
    df["Name"] = df["Kernel_Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
    v = df[df["Counter_Name"] == "SQ_INSTS_VALU"].groupby("Name")["Counter_Value"].sum().sort_values(ascending=False)
    (v / v.sum()).head(14).to_csv("$O/valu_share_d$d.csv")
    print("VALU share:"); print((v / v.sum()).head(12).round(3).to_string())
PY
rm -rf $O/stats_d$d $O/pmc_d$d
done
