"""Isolated timing of havoc_mi355x_rdoq on the 1080p workload's TU tables (coefficients made by tu_forward on the device).
usage: python profiles/rdoq_bench.py [reps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from turingcodec_amd.havoc import Havoc                       # noqa: E402
from turingcodec_amd.workload import FrameWorkload            # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hv = Havoc(stream="new")
wl = FrameWorkload(1920, 1080, 8)
dev = bench.DeviceFrame(hv, wl)
dev.step()
hv.sync()
out = {}
for name, fn in dev.launches:
    if name != "rdoq":
        continue
    fn()
    hv.timer_start()
    for _ in range(reps):
        fn()
    out.setdefault("ms", []).append(round(hv.timer_stop_ms() / reps, 4))
out["total_ms"] = round(sum(out["ms"]), 4)
out["groups"] = [f"{k}:{len(g['jobs'])}" for k, g in sorted(dev.tu.items(), reverse=True)]
print(json.dumps(out))
