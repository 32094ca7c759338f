"""Gantt summary of the overlapped step from a rocprofv3 --kernel-trace csv: a window of consecutive kernels (start offset,
duration, stream) and the share of time with k kernels running at once.   python profiles/timeline.py <kernel_trace.csv> [n]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "havoc_gpu" not in n:
        continue
    n = n.replace("void ", "").replace("havoc_gpu::", "").replace("(anonymous namespace)::", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Stream_Id", 0) or 0)))
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 70
mid = len(rows) // 2
win = rows[mid:mid + per]
t0 = win[0][0]
print("kernels in window", len(win), "span us", (max(e for _, e, _, _ in win) - t0) / 1e3)
for s, e, n, st in win:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  s{st:<3d} {n[:70]}")
ev = []
for s, e, _, _ in rows[len(rows) // 4: 3 * len(rows) // 4]:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {}
cur, last = 0, ev[0][0]
for t, d in ev:
    busy[cur] = busy.get(cur, 0) + (t - last)
    cur += d
    last = t
tot = sum(busy.values())
print("time share by number of kernels running concurrently:", {k: round(v / tot, 3) for k, v in sorted(busy.items())})
