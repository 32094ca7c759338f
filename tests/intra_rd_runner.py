"""Runs the intra RD-refinement batch client (libhavoc_search.so: havoc_search_intra_rd, turingcodec_amd/search/tu_search.cpp) on the intra
partitions of the workload (35-mode stage first, then every candidate mode of its refinement order reconstructed) against the same
decisions taken one candidate at a time through the reference's intra table, transform tables and Rdoq.cpp (tests/search_client.cpp:
client_intra_rd).  --device mock | real.  One JSON line."""
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
    ap.add_argument("--limit", type=int, default=0, help="partitions per size (0 = all of the workload's)")
    ap.add_argument("--expected", choices=["ref", "oracle"], default="ref")
    args = ap.parse_args()
    W, H = (int(v) for v in args.res.split("x"))
    BD = args.bit_depth
    S = 1 if BD == 8 else 2
    if args.device == "mock":
        C.CDLL(build_mock(), mode=C.RTLD_GLOBAL)
    from turingcodec_amd import decisions, workload
    wl = workload.FrameWorkload(W, H, BD, args.seed + 9, qp=args.qp)
    src = aligned(wl.luma)
    lam = workload.picture_lambda(args.qp, non_reference=False)
    rsl = st.reciprocal_sqrt_lambda(args.qp)
    quant = decisions.rqt_quant(args.qp, BD)
    states = wl.rdoq_states
    cx = (W + 63) // 64
    try:
        ref = st.Client("ref", 3) if args.expected == "ref" else st.Client("oracle")
        expected_from = "reference tables + Rdoq.cpp (oracle/_ref)" if args.expected == "ref" else "CPU oracle"
    except (FileNotFoundError, OSError):
        ref, expected_from = st.Client("oracle"), "CPU oracle"
    report = {"device": args.device, "res": args.res, "bit_depth": BD, "qp": args.qp, "expected_from": expected_from, "sizes": {}}

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
    d_src, d_states = up(src), up(np.ascontiguousarray(states))
    total_bad = 0
    groups, expected = [], {}
    for log2 in (2, 3, 4, 5):
        jobs = np.ascontiguousarray(wl.intra_search[log2])
        if args.limit:
            jobs = np.ascontiguousarray(jobs[:args.limit])
        n = len(jobs)
        if not n:
            continue
        nb = aligned(wl.intra_search_nb[log2])
        ictx = st.make_intra_contexts(n, log2, args.seed + log2)
        pos = jobs[:, 0] % wl.plane_len
        ctu = ((pos // wl.stride - 96) // 64) * cx + (pos % wl.stride - 96) // 64
        # stage 1 (35-mode SATD + order) through the checker, stage 2 both ways on that order
        order = ref.intra_order(ictx, rsl, ref.intra35(BD, log2, src, wl.stride, nb, jobs))
        t0 = time.perf_counter()
        exp, exp_rec = ref.intra_rd(BD, log2, src, wl.stride, nb, jobs, order, ictx, ctu, states, quant[log2 - 2], lam, 1.0 / lam)
        t_ref = time.perf_counter() - t0
        d_nb = up(nb)
        rec = np.zeros((n, 1 << 2 * log2), src.dtype)
        d_rec = up(rec)
        for attempt in range(2):
            t0 = time.perf_counter()
            got, stats = decisions.intra_rd(ctx, S, BD, log2, d_src, wl.stride, d_nb, jobs, order, ictx, ctu, d_states, quant[log2 - 2], lam, 1.0 / lam, d_rec)
            t_dev = time.perf_counter() - t0
        assert dev.havoc_mi355x_d2h(ctx, rec.ctypes.data, d_rec, rec.nbytes) == 0
        bad = int(sum(got[i].tobytes() != exp[i].tobytes() for i in range(n)))
        total_bad += bad + (0 if np.array_equal(rec, exp_rec) else 1)
        rec2 = np.zeros_like(rec)
        groups.append(dict(log2=log2, n=n, d_nb=d_nb, d_jobs=up(jobs), d_ictx=up(np.ascontiguousarray(ictx)), d_ctu=up(np.ascontiguousarray(ctu, np.int32)), d_rec=up(rec2)))
        expected[log2] = (exp, exp_rec)
        report["sizes"][str(1 << log2)] = {"partitions": n, "candidates": int(stats.candidates), "launches": int(stats.launches), "mismatching": bad,
                                           "reconstructions_equal": bool(np.array_equal(rec, exp_rec)), "seconds_batch": round(t_dev, 5),
                                           "seconds_per_call_one_core": round(t_ref, 4), "champion_is_first_candidate": float(np.mean(exp["index"] == 0)),
                                           "coded": float(np.mean(exp["outcome"]["cbf"] != 0))}
    # both stages with the decisions taken on the device (havoc_search_intra_device): all sizes in one call, the same champions
    for attempt in range(2):
        t0 = time.perf_counter()
        got, stats = decisions.intra_device(ctx, S, BD, d_src, wl.stride, groups, d_states, quant, rsl, lam, 1.0 / lam)
        t_dev = time.perf_counter() - t0
    bad = 0
    for g in groups:
        exp, exp_rec = expected[g["log2"]]
        rec = np.zeros_like(exp_rec)
        assert dev.havoc_mi355x_d2h(ctx, rec.ctypes.data, g["d_rec"], rec.nbytes) == 0
        bad += int(sum(got[g["log2"]][i].tobytes() != exp[i].tobytes() for i in range(g["n"]))) + (0 if np.array_equal(rec, exp_rec) else 1)
    report["device_decisions"] = {"mismatching": bad, "launches": int(stats.launches), "candidates": int(stats.candidates), "seconds": round(t_dev, 5),
                                  "candidates_per_call_arm": int(sum(v["candidates"] for v in report["sizes"].values()))}
    total_bad += bad
    report["mismatches"] = total_bad
    print(json.dumps(report))


if __name__ == "__main__":
    main()
