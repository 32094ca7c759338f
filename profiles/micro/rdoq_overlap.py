"""The five rdoq launches of a picture's TU chains side by side (one lane each, HIP graph), replayed -- for rocprofv3 --kernel-trace + profiles/timeline.py: do the walk
kernels of different transform sizes share the machine, and what does each last then?   python profiles/micro/rdoq_overlap.py [WxH [qp [reps]]]"""
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
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hv = Havoc(stream="new")
wl = FrameWorkload(int(res.split("x")[0]), int(res.split("x")[1]), 8, qp=qp)
dev = step.DeviceFrame(hv, wl)
dev.step()
hv.sync()
tu_chains = [ch for ch in dev.chains if dev.launches[ch[0]][0] == "tu_forward"]
sel = [[i for i in ch if dev.launches[i][0] == "rdoq"] for ch in tu_chains]
sel = [s for s in sel if s]


def forked():
    hv.fork(len(sel))
    for k, s in enumerate(sel):
        hv.lane(k)
        for i in s:
            dev.launches[i][1]()
    hv.join()


g = hv.graph_capture(forked)
hv.graph_launch(g)
hv.sync()
hv.timer_start()
for _ in range(reps):
    hv.graph_launch(g)
ms = hv.timer_stop_ms() / reps
print(json.dumps({"res": res, "qp": qp, "rdoq_side_by_side_ms": round(ms, 4)}))
