"""Isolated timing of the SAO primitives on a 1920x1080 8-bit picture: one statistics job and one filter job per CTU and colour
component (510 luma + 2 x 510 chroma blocks).   python profiles/sao_bench.py [reps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                   # noqa: E402
from turingcodec_amd import havoc as hm                        # noqa: E402
from turingcodec_amd.havoc import Havoc, SAO_JOB_DT            # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
hv = Havoc(stream="new")
rng = np.random.default_rng(1)
out = {}
for name, W, H, pad, ctu in (("luma", 1920, 1080, 96, 64), ("chroma (one plane)", 960, 540, 48, 32)):
    stride = (W + 2 * pad + 63) & ~63
    rec = rng.integers(0, 256, (H + 2 * pad) * stride).astype(np.uint8)
    src = np.clip(rec.astype(int) + rng.integers(-4, 5, rec.shape), 0, 255).astype(np.uint8)
    rects = [(x, y, min(ctu, W - x), min(ctu, H - y)) for y in range(0, H, ctu) for x in range(0, W, ctu)]
    sj = np.array([[(y + pad) * stride + x + pad] * 2 + [w, h] for x, y, w, h in rects], np.int32)
    fj = np.zeros(len(rects), SAO_JOB_DT)
    for i, (x, y, w, h) in enumerate(rects):
        offs = np.zeros(32, np.int16)
        offs[1:5] = (3, 1, -1, -3)
        fj[i] = (sj[i, 0], sj[i, 0], w, h, 2, i % 4, offs, (0, 0))
    d_src, d_rec, d_sj = hv.up(src), hv.up(rec), hv.up(sj)
    d_dst = hv.zeros(len(rec), np.uint8)
    with torch.cuda.stream(hv.tstream):
        d_stats = torch.zeros(105 * len(rects), dtype=torch.int64, device=hv.device)
        d_fj = torch.from_numpy(fj.view(np.uint8).reshape(-1)).to(hv.device)
    for what, fn in (("stats", lambda: hv._ck(hv.L.havoc_mi355x_sao_stats(hv.h, 1, 8, hm._ptr(d_src), stride, hm._ptr(d_rec), stride, hm._ptr(d_sj), len(rects), hm._ptr(d_stats)))),
                     ("filter", lambda: hv._ck(hv.L.havoc_mi355x_sao_filter(hv.h, 1, 8, hm._ptr(d_dst), stride, hm._ptr(d_rec), stride, hm._ptr(d_fj), len(rects))))):
        fn()
        hv.timer_start()
        for _ in range(reps):
            fn()
        out[f"{name} {what} us"] = round(hv.timer_stop_ms() / reps * 1e3, 2)
print(json.dumps(out))
