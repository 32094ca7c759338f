"""Isolated timing of havoc_mi355x_rdoq on the 1080p workload's TU tables (coefficients made by tu_forward on the device).
usage: python profiles/rdoq_bench.py [reps [WxH [qp]]]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turingcodec_amd import step                              # noqa: E402
from turingcodec_amd.havoc import Havoc                       # noqa: E402
from turingcodec_amd.workload import FrameWorkload            # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hv = Havoc(stream="new")
res = sys.argv[2] if len(sys.argv) > 2 else "1920x1080"
qp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
wl = FrameWorkload(int(res.split("x")[0]), int(res.split("x")[1]), 8, qp=qp)
dev = step.DeviceFrame(hv, wl)
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
out["res"], out["qp"] = res, qp
# how many groups a block of each size makes the walk visit (the scan's view): mean and maximum
import torch
for key, g in sorted(dev.tu.items(), reverse=True):
    n2 = g["n"] ** 2
    c = hv.down(g["coef"], np.int16).reshape(-1, g["n"] // 4 if g["n"] > 4 else 1, 4, g["n"] // 4 if g["n"] > 4 else 1, 4) if g["n"] > 4 else hv.down(g["coef"], np.int16).reshape(-1, 1, 4, 1, 4)
    from turingcodec_amd.workload import quant_params
    qs, sh, _ = quant_params(qp, key[0], 8, False)
    nzg = ((np.abs(c.astype(np.int64)) * qs + (1 << (sh - 1))) >> sh).max(axis=(2, 4)) > 0
    cnt = nzg.reshape(len(c), -1).sum(1)
    out.setdefault("groups_to_walk", {})[str(key)] = {"mean": round(float(cnt.mean()), 2), "max": int(cnt.max()), "p99": int(np.percentile(cnt, 99))}
print(json.dumps(out))
