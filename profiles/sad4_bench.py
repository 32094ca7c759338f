#!/usr/bin/env python3
"""The 4-way SAD calls of one 1080p picture (1.27 M calls, the bench's workload) timed alone: `calls` = havoc_mi355x_sad4 (a window per call, k_sad4w),
`runs` = havoc_mi355x_sad4_runs (a window per search, k_sad4r; HAVOC_SAD4_RUN_WAVES picks the workgroup size, HAVOC_SAD4_RUN_UNROLL=1|2 the round's first form (16 lanes per
call) instead of a lane per candidate, HAVOC_SAD4_RUN_SRC=l the source block from its LDS copy instead of scalar loads; each read once per process).
HAVOC_SAD4_RES=WxHxBITS picks the picture (default 1920x1080x8).
    python profiles/sad4_bench.py runs|calls [reps]
Prints one JSON line; run under rocprofv3 for the counters (profiles/gpu_call_r05b.sh)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turingcodec_amd import Havoc  # noqa: E402
from turingcodec_amd.workload import FrameWorkload  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "runs"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hv = Havoc(0, stream="new")
res = os.environ.get("HAVOC_SAD4_RES", "1920x1080x8").split("x")      # width x height x bit depth
wl = FrameWorkload(int(res[0]), int(res[1]), int(res[2]), 11)
luma, jobs = hv.up(wl.luma), hv.up(wl.sad4)
boxed = os.environ.get("HAVOC_SAD4_BOX", "1") == "1"      # 0: runs without a box (the kernel reduces every run's box itself)
runs = Havoc.sad4_make_runs(wl.sad4, int(os.environ.get("HAVOC_SAD4_MAX_RUN", "0")), wl.stride if boxed else None, wl.S)
policy = os.environ.get("HAVOC_SAD4_POLICY")      # "64:16,32:32,16:64,8:128": calls per run by block width (experiment: work per workgroup evened out); runs without a box
if policy:
    cap = {int(k): int(v) for k, v in (kv.split(":") for kv in policy.split(","))}
    out_runs = []
    for first, count in Havoc.sad4_make_runs(wl.sad4, 128)[:, :2]:
        c = cap.get(int(wl.sad4[first, 5]), 128)
        for b in range(0, count, c):
            out_runs.append((first + b, min(c, count - b)))
    runs = Havoc.as_runs(np.array(out_runs, np.int32))
d_runs = hv.up(runs)
out = hv.zeros(4 * len(wl.sad4), np.int32)
fn = (lambda: hv.sad4_runs_d(luma, wl.stride, luma, wl.stride, jobs, d_runs, out)) if form == "runs" else (lambda: hv.sad4_d(luma, wl.stride, luma, wl.stride, jobs, out))
fn()
hv.sync()
best = 1e9
for _ in range(3):
    hv.timer_start()
    for _ in range(reps):
        fn()
    best = min(best, hv.timer_stop_ms() / reps)
size = {}
for w in (8, 16, 32, 64):
    size[w] = int((wl.sad4[runs[:, 0], 5] == w).sum())
print(json.dumps({"form": form, "picture": "x".join(res), "waves": os.environ.get("HAVOC_SAD4_RUN_WAVES", "4"), "unroll": os.environ.get("HAVOC_SAD4_RUN_UNROLL", "0"), "src": os.environ.get("HAVOC_SAD4_RUN_SRC", "scalar"), "caps": os.environ.get("HAVOC_SAD4_CAPS", "16,48,128"), "policy": policy, "max_run": os.environ.get("HAVOC_SAD4_MAX_RUN", "0"), "boxed": bool(boxed and not policy), "ms": round(best, 4), "calls": int(len(wl.sad4)), "runs": int(len(runs)), "runs_by_width": size,
                  "checksum": int(hv.down(out, np.int32).astype(np.int64).sum())}))
