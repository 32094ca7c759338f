#!/bin/bash
# round 4, call v: kernel timeline of 8 pictures in flight (decision step): where a picture's post-search launches wait
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04v; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - <<PY > $O/analysis.txt 2>&1
import pandas as pd, numpy as np
d=pd.read_csv("$f")
print(d.columns.tolist()); print(len(d))
d["name"]=d["Kernel_Name"].str.replace("void ","").str.replace("havoc_gpu::","").str.replace("(anonymous namespace)::","",regex=False).str.split("(").str[0].str.slice(0,40)
d["dur"]=d["End_Timestamp"]-d["Start_Timestamp"]
t1=d["End_Timestamp"].max()
w=d[d["Start_Timestamp"]>t1-1.0e9]           # the last second: 8 in flight
qcol=[c for c in d.columns if "Queue" in c or "Stream" in c]
print(qcol)
for q in qcol: print(q, w[q].nunique())
q=qcol[0]
rows=[]
for qi,g in w.groupby(q):
    g=g.sort_values("Start_Timestamp")
    s=g[g["name"].str.contains("k_search_rows")]
    if len(s)<3: continue
    # per step: from a search_rows end to the next search_rows start
    ends=s["End_Timestamp"].values[:-1]; starts=s["Start_Timestamp"].values[1:]
    gap=(starts-ends)/1e6
    busy=[]
    for a,b in zip(ends,starts):
        k=g[(g["Start_Timestamp"]>=a)&(g["End_Timestamp"]<=b)]
        busy.append(k["dur"].sum()/1e6)
    rows.append((qi,len(s),s["dur"].mean()/1e6,np.mean(gap),np.mean(busy),len(g)/max(1,len(s))))
print("queue, steps, search_ms, between_searches_ms, kernels_busy_in_between_ms, launches_per_step")
for r in rows: print(r)
# the gaps of one queue's timeline inside one step
g=w[w[q]==sorted(w[q].unique())[2]].sort_values("Start_Timestamp")
s=g[g["name"].str.contains("k_search_rows")]
a,b=s["End_Timestamp"].values[3], s["Start_Timestamp"].values[4]
k=g[(g["Start_Timestamp"]>=a)&(g["End_Timestamp"]<=b)]
prev_end=a; prev="k_search_rows"
print("one step of one queue: gaps > 80 us (after kernel, gap_us, before kernel, at_ms)")
tot=0
for _,r in k.iterrows():
    gap=(r["Start_Timestamp"]-prev_end)/1e3
    if gap>80: print(prev, round(gap), r["name"], round((r["Start_Timestamp"]-a)/1e6,2)); tot+=gap
    prev_end=r["End_Timestamp"]; prev=r["name"]
print("last kernel -> next search", round((b-prev_end)/1e3), "sum of listed gaps us", round(tot), "step in-between ms", (b-a)/1e6)
# which kernels take the in-between time (sum of durations per step, top 15)
ws=w.groupby("name")["dur"].agg(["sum","count","mean"]).sort_values("sum",ascending=False).head(25)
ws["sum_ms"]=ws["sum"]/1e6; ws["mean_us"]=ws["mean"]/1e3
print(ws[["sum_ms","count","mean_us"]].to_string())
PY
rm -rf $O/tr
head -c 6000 $O/analysis.txt
