import sys, os, numpy as np
sys.path.insert(0, ".")
lib = sys.argv[1]
import turingcodec_amd.havoc as H
if lib != "base":
    H.LIB_PATH = os.path.abspath(lib)
from turingcodec_amd import Havoc
from turingcodec_amd.workload import FrameWorkload
hv = Havoc(0, stream="new")
wl = FrameWorkload(1920, 1080, 8, 11)
tot = 0
for log2 in (2, 3, 4, 5):
    j = wl.intra[log2]
    n = 1 << log2
    dj, nb = hv.up(j), hv.up(wl.intra_nb[log2])
    dst = hv.zeros(len(j) << (2 * log2), np.uint8)
    run = lambda: hv.intra_d(8, log2, dst, n, nb, dj)
    run(); hv.sync()
    best = 1e9
    for _ in range(3):
        hv.timer_start()
        for _ in range(5): run()
        best = min(best, hv.timer_stop_ms() / 5)
    tot += best
    print(lib, "log2", log2, f"{best*1e3:.1f} us", int(dst.to(__import__('torch').int64).sum().item()))
print(lib, "total", f"{tot*1e3:.1f}")
