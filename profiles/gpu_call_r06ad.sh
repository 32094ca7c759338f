#!/bin/bash
# GPU call r06ad: kernel trace of an intra picture's chain (1080p): which kernels fill a level's ~160 us
tag=${1:-r06ad}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/intra_once.py <<PY
import sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from turingcodec_amd.decisions import IntraChainPicture
from turingcodec_amd.havoc import Havoc
hv = Havoc(stream="new")
ip = IntraChainPicture(hv, 1920, 1080, 8, 32, seed=17)
ip.step()
t0 = time.perf_counter(); ip.step(); print("seconds", time.perf_counter() - t0, "levels", ip.nlevels, "launches", ip.launches)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python /tmp/intra_once.py > $O/run.log 2>&1; tail -2 $O/run.log | cut -c1-200
python - <<PY
import glob, pandas as pd
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)
st = pd.read_csv(f[0]); st = st[st["Name"].str.contains("havoc_gpu")].copy()
st["Name"] = st["Name"].str.replace("void ", "").str.replace("havoc_gpu::", "").str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
st = st.sort_values("TotalDurationNs", ascending=False)
st.to_csv("$O/intra_chain_kernel_stats.csv", index=False)
print(st[["Name", "Calls", "TotalDurationNs", "AverageNs", "MaxNs"]].head(24).to_string())
print("sum of kernel time ms", st.TotalDurationNs.sum() / 1e6, "calls", st.Calls.sum())
t = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)
tr = pd.read_csv(t[0]); tr = tr[tr["Kernel_Name"].str.contains("havoc_gpu")].sort_values("Start_Timestamp")
half = tr.iloc[len(tr) // 2:]      # the second (timed) step
span = (half.End_Timestamp.max() - half.Start_Timestamp.min()) / 1e6
ev = sorted([(s, 1) for s in half.Start_Timestamp] + [(e, -1) for e in half.End_Timestamp])
busy = {}; cur = 0; last = ev[0][0]
for tt, d in ev:
    busy[cur] = busy.get(cur, 0) + (tt - last); cur += d; last = tt
tot = sum(busy.values())
print("second step: span ms", span, "kernels", len(half), "share of time with k kernels running:", {k: round(v / tot, 3) for k, v in sorted(busy.items())})
PY
rm -rf $O/trace
