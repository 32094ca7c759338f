"""Runs the residual-quadtree batch client (libhavoc_search.so: havoc_search_rqt, turingcodec_amd/search/tu_search.cpp) against the same
decisions taken one block at a time through the reference's tables + Rdoq.cpp (tests/search_client.cpp: client_rqt), in a subprocess of
the tests.  --device mock: tests/mock_device.c stands in for libhavoc_mi355x.so; --device real: the MI355X library.  One JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import search_tools as st  # noqa: E402
from search_runner import aligned, build_mock  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", choices=["mock", "real"], default="real")
    ap.add_argument("--res", default="416x240")
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--expected", choices=["ref", "oracle"], default="ref")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    W, H = (int(v) for v in args.res.split("x"))
    BD = args.bit_depth
    S = 1 if BD == 8 else 2
    if args.device == "mock":
        C.CDLL(build_mock(), mode=C.RTLD_GLOBAL)
    from turingcodec_amd import decisions, workload
    planes, stride = st.clip_planes(W, H, args.seed + 4, BD)
    src, ref0 = aligned(planes[0]), planes[1]
    pad = 96
    # the units' prediction: the list-0 picture displaced by the clip's motion plus a little (a plane of its own, stride W)
    r2 = ref0.reshape(-1, stride)
    pred = aligned(np.ascontiguousarray(r2[pad - 2:pad - 2 + H, pad - 3:pad - 3 + W]).ravel())
    cx = (W + 63) // 64
    cus = decisions.rqt_units(W, H, cx)
    quant = decisions.rqt_quant(args.qp, BD)
    lam = workload.picture_lambda(args.qp)
    rng = np.random.default_rng(args.seed)
    states = np.clip(rng.integers(4, 100, 128)[None, :] + rng.integers(-6, 7, (cx * ((H + 63) // 64), 128)), 0, 125).astype(np.uint8)
    report = {"device": args.device, "res": args.res, "bit_depth": BD, "qp": args.qp, "units": int(len(cus))}
    try:
        ref = st.Client("ref", 3) if args.expected == "ref" else st.Client("oracle")
        report["expected_from"] = "reference tables + Rdoq.cpp (oracle/_ref)" if args.expected == "ref" else "CPU oracle"
    except (FileNotFoundError, OSError):
        ref = st.Client("oracle")
        report["expected_from"] = "CPU oracle"
    t0 = time.perf_counter()
    exp, exp_rec = ref.rqt(BD, src, stride, pad, pred, W, states, quant, lam, 1.0 / lam, cus)
    report["expected_seconds"] = round(time.perf_counter() - t0, 4)

    dev = C.CDLL(os.path.join(st.BUILD, "mock", "libhavoc_mi355x.so") if args.device == "mock" else os.path.join(ROOT, "turingcodec_amd", "libhavoc_mi355x.so"),
                 mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    dev.havoc_mi355x_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    dev.havoc_mi355x_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    dev.havoc_mi355x_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    dev.havoc_mi355x_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    dev.havoc_mi355x_last_error.restype = C.c_char_p
    ctx = vp()
    assert dev.havoc_mi355x_create(C.byref(ctx), 0, vp(-1 & 0xFFFFFFFFFFFFFFFF)) == 0, dev.havoc_mi355x_last_error()

    def up(a):
        d = vp()
        assert dev.havoc_mi355x_malloc(ctx, C.byref(d), a.nbytes + 256) == 0
        assert dev.havoc_mi355x_h2d(ctx, d, a.ctypes.data, a.nbytes) == 0
        return d
    d_src, d_pred, d_states = up(src), up(pred), up(states)
    rec = np.zeros_like(src)
    d_rec = up(rec)
    origin = pad * stride + pad
    for attempt in range(args.repeat):
        assert dev.havoc_mi355x_h2d(ctx, d_rec, rec.ctypes.data, rec.nbytes) == 0
        t0 = time.perf_counter()
        got, stats = decisions.rqt(ctx, S, BD, d_src, origin, stride, d_pred, W, d_rec, origin, stride, d_states, quant, lam, 1.0 / lam, cus)
        t = time.perf_counter() - t0
    got_rec = np.zeros_like(src)
    assert dev.havoc_mi355x_d2h(ctx, got_rec.ctypes.data, d_rec, got_rec.nbytes) == 0
    d = stats.as_dict()
    d.update({"seconds": round(t, 5), "units_per_second": round(len(cus) / t, 1), "launches_per_ctu": round(d["launches"] / (cx * ((H + 63) // 64)), 4)})
    for k in ("seconds_gpu", "seconds_host", "seconds_total"):
        d[k] = round(d[k], 5)
    report["rqt"] = d
    report["mismatching_units"] = int((got.tobytes() != exp.tobytes()) and sum(got[i].tobytes() != exp[i].tobytes() for i in range(len(got))))
    report["reconstruction_equal"] = bool(np.array_equal(got_rec, exp_rec))
    report["depth_histogram"] = {"depth0_tried": int(((exp["depth"] == 0) & (exp["tried_zero"] == 1)).sum()), "depth1": int((exp["depth"] == 1).sum()),
                                 "uncoded": int((exp["tried_zero"] == 0).sum())}
    print(json.dumps(report))


if __name__ == "__main__":
    main()
