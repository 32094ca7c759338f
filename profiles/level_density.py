#!/usr/bin/env python3
"""How dense the quantised levels of the bench workload's transform units are (residual -> forward transform -> Rdoq, CPU oracle, a sample per size),
as shares by class: 0 = no level, k = 2^(k-1) .. 2^k - 1 levels -- to hold against the reference encoder's own density, which profiles/measure_call_mix.py
tallies from its de-quantiser calls ("dequant_nonzero <side>x<class>" in profiles/r04_reference_call_mix_1080p.json).  RDOQ's cost on CPU and GPU
depends on it, so the workload must be in the measured regime (it is: see the end of this file's output)."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from turingcodec_amd import workload as W
from turingcodec_amd.workload import FrameWorkload, quant_params, dequant_params, PAD
from reflibs import Oracle
o = Oracle()
def hist(wl, nsamp=1500, seed=1):
    rng = np.random.default_rng(seed)
    out = {}
    for (log2, tr), g in wl.tu.items():
        n = g["n"]; jobs = g["jobs"]; src4 = g["src"]
        idx = rng.choice(len(jobs), min(nsamp, len(jobs)), replace=False)
        qs, qsh, _ = quant_params(wl.qp, log2, wl.bit_depth, False); inv,_ = dequant_params(wl.qp, log2, wl.bit_depth)
        h = np.zeros(8, int)
        for i in idx:
            so, po = int(src4[i,0]), int(src4[i,1])
            st = wl.stride
            sblk = np.stack([wl.luma[so + r*st: so + r*st + n] for r in range(n)]).astype(np.int32)
            pblk = np.stack([wl.luma[po + r*st: po + r*st + n] for r in range(n)]).astype(np.int32)
            res = np.ascontiguousarray((sblk - pblk).astype(np.int16).ravel())
            coef = np.zeros(n*n, np.int16)
            o.transform(coef, 0, res, 0, n, log2, tr, wl.bit_depth)
            lv, _ = o.rdoq(coef, log2, 0, int(g["scan_idx"][i]), int(g["is_intra"][i]), 1, qs, qsh, inv, wl.bit_depth, wl.rdoq_lambda, wl.rdoq_states[g["ctx_index"][i]])
            nz = int(np.count_nonzero(lv)); c = 0
            while (1 << c) <= nz: c += 1
            h[min(c,7)] += 1
        out[(log2,tr)] = np.round(h / h.sum(), 3)
    return out
if __name__ == "__main__":
    wl = FrameWorkload(1920,1080,8,11)
    for k,v in sorted(hist(wl).items()): print("workload", k, v)
    import json
    m = json.load(open(os.path.join(ROOT, "profiles", "r04_reference_call_mix_1080p.json")))["by_size"]["dequant_nonzero"]
    for side in (4, 8, 16, 32):
        h = np.array([m.get(f"{side}x{c}", 0) for c in range(8)], float)
        print("reference encoder", f"{side}x{side}", np.round(h / h.sum(), 3))
