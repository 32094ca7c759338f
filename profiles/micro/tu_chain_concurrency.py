"""How long do the TU chain's launch groups take when the five transform sizes run SIDE BY SIDE (fork/join lanes, one size per lane, HIP graph) instead of one after
the other?  An upper bound for what one launch covering all size classes would take.   python profiles/micro/tu_chain_concurrency.py [WxH [qp]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from turingcodec_amd import step                              # noqa: E402
from turingcodec_amd.havoc import Havoc                       # noqa: E402
from turingcodec_amd.workload import FrameWorkload            # noqa: E402

res = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hv = Havoc(stream="new")
wl = FrameWorkload(int(res.split("x")[0]), int(res.split("x")[1]), 8, qp=qp)
dev = step.DeviceFrame(hv, wl)
dev.step()
hv.sync()
# the TU chains: lists of launch indices whose first launch is a tu_forward
tu_chains = [ch for ch in dev.chains if dev.launches[ch[0]][0] == "tu_forward"]
out = {"res": res, "qp": qp, "chains": len(tu_chains)}


def timed(fn, reps=20):
    fn()
    hv.sync()
    best = 1e9
    for _ in range(3):
        hv.timer_start()
        for _ in range(reps):
            fn()
        best = min(best, hv.timer_stop_ms() / reps)
    return round(best, 4)


for group in ("tu_forward", "rdoq", "tu_reconstruct", "ssd"):
    sel = [[i for i in ch if dev.launches[i][0] == group] for ch in tu_chains]
    sel = [s for s in sel if s]
    serial = lambda: [dev.launches[i][1]() for s in sel for i in s]

    def forked():
        hv.fork(len(sel))
        for k, s in enumerate(sel):
            hv.lane(k)
            for i in s:
                dev.launches[i][1]()
        hv.join()
    g = hv.graph_capture(forked)
    out[group] = {"serial_ms": timed(serial), "side_by_side_ms": timed(lambda: hv.graph_launch(g))}
# the whole chains side by side, and one after the other
serial = lambda: [dev.launches[i][1]() for ch in tu_chains for i in ch]


def forked_all():
    hv.fork(len(tu_chains))
    for k, ch in enumerate(tu_chains):
        hv.lane(k)
        for i in ch:
            dev.launches[i][1]()
    hv.join()
g = hv.graph_capture(forked_all)
out["whole_chains"] = {"serial_ms": timed(serial), "side_by_side_ms": timed(lambda: hv.graph_launch(g))}
print(json.dumps(out))
