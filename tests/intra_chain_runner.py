"""Runs libhavoc_search.so's havoc_search_intra_chain (an INTRA picture with the real dependencies between its partitions: turingcodec_amd/search/tu_search.cpp) over the tables
turingcodec_amd.decisions.intra_chain_tables builds, and the reference's block-at-a-time loop (tests/test_intra_chain.py: _host_chain, through the reference's intra / Hadamard /
transform tables + Rdoq.cpp) on the same picture.  --device mock: tests/mock_device.c stands in for libhavoc_mi355x.so (the client's level loop, slices and slot handling on the
CPU, with the chain's device-side steps restated on the host); --device real: the MI355X library.  One JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time
from types import SimpleNamespace

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
    ap.add_argument("--seed", type=int, default=17)
    ap.add_argument("--qp", type=int, default=32)
    args = ap.parse_args()
    W, H = (int(v) for v in args.res.split("x"))
    BD, PAD = args.bit_depth, 96
    S = 1 if BD == 8 else 2
    dt = np.uint8 if S == 1 else np.uint16
    if args.device == "mock":
        C.CDLL(build_mock(), mode=C.RTLD_GLOBAL)
    from turingcodec_amd import decisions, workload
    from test_intra_chain import _host_chain
    plane = workload.pad_plane(workload.synth_frames(W, H, 1, args.seed, BD)[0][0], PAD)
    stride = plane.shape[1]
    src = aligned(np.ascontiguousarray(plane.ravel()))
    t = decisions.intra_chain_tables(W, H, BD, args.qp, args.seed, stride, PAD)

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
        a = np.ascontiguousarray(a)
        d = vp()
        assert dev.havoc_mi355x_malloc(ctx, C.byref(d), a.nbytes + 256) == 0
        assert dev.havoc_mi355x_h2d(ctx, d, a.ctypes.data, a.nbytes) == 0
        return d

    def down(d, shape, dtype):
        a = np.zeros(shape, dtype)
        assert dev.havoc_mi355x_d2h(ctx, a.ctypes.data, d, a.nbytes) == 0
        return a
    d_src, d_rec = up(src), up(np.zeros_like(src))
    d_owner, d_modes = up(t["owner"].ravel()), up(np.zeros(t["owner"].size, np.uint8))
    d_states = up(t["rdoq_states"])
    layout = (C.c_int32 * 8)(W, H, stride, PAD, t["owner"].shape[1], BD, 6, 1)      # strong_intra_smoothing: the reference encoder's default
    table = np.zeros(len(t["sizes"]), decisions.INTRA_CHAIN_SIZE_DT)
    keep = []
    for row, (log2, g) in zip(table, t["sizes"].items()):
        m, nn = g["m"], g["nn"]
        g["best"] = np.zeros(m, decisions.INTRA_RD_RESULT_DT)
        g["d_ictx"], g["d_blocks"] = up(g["ictx"]), up(np.zeros(m * nn * nn, dt))
        f = np.ascontiguousarray(g["first"], np.int32)
        keep.append(f)
        row["log2"], row["n"] = log2, m
        row["d_neighbours"], row["d_jobs"], row["d_ictx"], row["d_ctx_index"] = up(np.zeros(m * 2 * (4 * nn + 1), dt)).value, up(g["jobs"]).value, g["d_ictx"].value, up(g["ctu"]).value
        row["d_parts"], row["d_blocks"], row["first"], row["out"] = up(g["chain"]).value, g["d_blocks"].value, f.ctypes.data, g["best"].ctypes.data
    quant = np.ascontiguousarray(t["quant"], np.int32)
    stats = decisions.RqtStats()
    lam = t["lam"]
    t0 = time.perf_counter()
    rc = decisions.lib().havoc_search_intra_chain(ctx, S, BD, layout, d_src, stride, d_rec, d_owner, d_modes, table.ctypes.data, len(table), t["nlevels"], d_states,
                                                  quant.ctypes.data, t["rsl"], float(lam), 1.0 / lam, 1, C.byref(stats))
    seconds = time.perf_counter() - t0
    assert rc == 0, (rc, dev.havoc_mi355x_last_error())
    parts = t["parts"]
    best = np.zeros(len(parts), decisions.INTRA_RD_RESULT_DT)
    cand = np.zeros((len(parts), 3), np.int32)
    for g in t["sizes"].values():
        best[g["sel"]] = g["best"]
        cand[g["sel"]] = down(g["d_ictx"], g["m"], decisions.INTRA_CTX_DT)["cand_mode_list"]
    rec = down(d_rec, src.shape, dt)
    # the reference's loop on the same picture
    ip = SimpleNamespace(W=W, H=H, PAD=PAD, stride=stride, bd=BD, host_src=src, owner=t["owner"], parts=parts, sizes=t["sizes"], rsl=t["rsl"], rdoq_states=t["rdoq_states"],
                         quant=t["quant"], lam=lam, dt=dt)
    try:
        ref, expected_from = st.Client("ref", 3), "reference tables + Rdoq.cpp (oracle/_ref)"
    except (FileNotFoundError, OSError):
        ref, expected_from = st.Client("oracle"), "CPU oracle"
    t0 = time.perf_counter()
    exp_best, exp_cand, exp_rec = _host_chain(ref, ip)
    t_ref = time.perf_counter() - t0
    g2d, e2d = rec.reshape(-1, stride), exp_rec.reshape(-1, stride)
    report = {"device": args.device, "res": args.res, "bit_depth": BD, "qp": args.qp, "expected_from": expected_from, "partitions": int(len(parts)), "levels": int(t["nlevels"]),
              "launches": int(stats.launches), "seconds_chain": round(seconds, 4), "seconds_reference_loop_one_core": round(t_ref, 3),
              "mismatching_champions": int(sum(best[i].tobytes() != exp_best[i].tobytes() for i in range(len(parts)))),
              "mismatching_cand_mode_lists": int(np.count_nonzero(np.any(cand != exp_cand, axis=1))),
              "reconstruction_equal": bool(np.array_equal(g2d[PAD:PAD + H, PAD:PAD + W], e2d[PAD:PAD + H, PAD:PAD + W])),
              "distinct_champion_modes": int(len(np.unique(best["mode"]))), "coded": float(np.mean(best["outcome"]["cbf"] != 0))}
    print(json.dumps(report))


if __name__ == "__main__":
    main()
