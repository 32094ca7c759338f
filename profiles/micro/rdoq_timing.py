"""Where a wavefront of the RDOQ walk kernels spends its cycles (diagnostic build: csrc/kernels_rdoq.hip with -DHAVOC_RDOQ_TIMING linked into
profiles/micro/libhavoc_mi355x_timing.so; `bash profiles/micro/build_timing.sh`).  Run with HAVOC_MI355X_LIB pointing at that library:
    HAVOC_MI355X_LIB=profiles/micro/libhavoc_mi355x_timing.so python profiles/micro/rdoq_timing.py [WxH [qp]]
Prints, per rdoq launch of the bench's TU tables, shader-clock cycles per wavefront by section."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from turingcodec_amd import step                              # noqa: E402
from turingcodec_amd import havoc as H                        # noqa: E402
from turingcodec_amd.workload import FrameWorkload            # noqa: E402

NAMES = ["stage_in", "before_loop", "pick_or_hop", "load_group", "prologue+loop_Z", "loop_B", "epilogue", "exchange+barrier", "replay", "finish_group", "end_barrier", "verdict"]
res = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hv = H.Havoc(stream="new")
L = C.CDLL(H.LIB_PATH)
buf = (C.c_ulonglong * 32)()
wl = FrameWorkload(int(res.split("x")[0]), int(res.split("x")[1]), 8, qp=qp)
dev = step.DeviceFrame(hv, wl)
dev.step()
hv.sync()
out = []
keys = [k for k, _ in sorted(dev.tu.items(), reverse=True)]
i = 0
for name, fn in dev.launches:
    if name != "rdoq":
        continue
    assert L.havoc_mi355x_debug_rdoq_timing(buf, 1) == 0
    fn()
    assert L.havoc_mi355x_debug_rdoq_timing(buf, 1) == 0
    v = list(buf)
    for base, kind, waves in ((0, "walk", v[15]), (16, "diag", v[31])):
        if not waves:
            continue
        per = {NAMES[k]: round(v[base + k] / waves) for k in range(12) if v[base + k]}
        out.append({"tu": str(keys[i]), "kernel": kind, "wavefronts": waves, "cycles_per_wavefront": sum(per.values()), "by_section": per})
    i += 1
for o in out:
    print(json.dumps(o))
