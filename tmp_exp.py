import sys, shutil, os, numpy as np
sys.path.insert(0, ".")
lib = sys.argv[1]
import turingcodec_amd.havoc as H
if lib != "base":
    H.LIB_PATH = os.path.abspath(lib)
from turingcodec_amd import Havoc
from turingcodec_amd.workload import FrameWorkload
hv = Havoc(0, stream="new")
wl = FrameWorkload(1920, 1080, 8, 11)
luma = hv.up(wl.luma)
for log2 in (3, 4, 5):
    j = wl.intra_search[log2]
    dj, nb = hv.up(j), hv.up(wl.intra_search_nb[log2])
    cost = hv.zeros(35 * len(j), np.int32)
    run = lambda: hv.intra_satd35_d(8, log2, luma, wl.stride, nb, dj, cost)
    run(); hv.sync()
    best = 1e9
    for _ in range(3):
        hv.timer_start()
        for _ in range(5): run()
        best = min(best, hv.timer_stop_ms() / 5)
    print(lib, "log2", log2, f"{best*1e3:.1f} us")
