#!/usr/bin/env python3
"""Runs the decision-loop clients against each other in ONE fresh process and prints a JSON report (tests/test_search.py
asserts on it; profiles/ keeps the GPU box's report).

    expected    tests/search_client.cpp over the reference's own havoc library (oracle/_ref): C tables
    classic     the same client over libhavoc_classic.so with the pictures registered (precompute and serve)
    batch       libhavoc_search.so, the batch client of libhavoc_mi355x.so

--device mock : a CPU stand-in for libhavoc_mi355x.so (tests/mock_device.c) is loaded first -- host logic only, no GPU;
--device real : the real library on a gfx950 device.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import search_tools as st  # noqa: E402


def build_mock():
    out = os.path.join(st.BUILD, "mock", "libhavoc_mi355x.so")
    srcs = [os.path.join(HERE, "mock_device.c"), os.path.join(ROOT, "oracle", "havoc_oracle.c"), os.path.join(ROOT, "oracle", "rdoq_oracle.c")]
    if not (os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-w", "-Wl,-soname,libhavoc_mi355x.so"] + srcs + ["-o", out])
    return out


def aligned(a, align=64):
    raw = np.empty(a.nbytes + align, np.uint8)
    o = (-raw.ctypes.data) % align
    out = raw[o:o + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


class Stats(C.Structure):
    _fields_ = [("rounds", C.c_int32), ("launches", C.c_int32), ("surfaces_small", C.c_int32), ("surfaces_large", C.c_int32),
                ("satd_jobs", C.c_int32), ("replays", C.c_int32), ("bytes_down", C.c_int64), ("seconds_gpu", C.c_double),
                ("seconds_host", C.c_double), ("seconds_total", C.c_double)]


FIELDS = ["mv", "mvd", "mv_integer", "mvp_flag", "wrote_2Nx2N", "calls", "cost_integer", "cost_subpel", "cost_mvd_zero"]


def same(a, b, fields=FIELDS):
    if len(a) == 0 and len(b) == 0:
        return []
    return [int(i) for i in np.flatnonzero(~np.all([np.all(a[f].reshape(len(a), -1) == b[f].reshape(len(b), -1), axis=1) for f in fields], axis=0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", choices=["mock", "real"], default="real")
    ap.add_argument("--res", default="640x360")
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--searches", type=int, default=300)
    ap.add_argument("--bi", type=int, default=40)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--skip", default="", help="comma list of: classic, batch, intra")
    args = ap.parse_args()
    W, H = (int(v) for v in args.res.split("x"))
    S = 1 if args.bit_depth == 8 else 2
    if args.device == "mock":
        C.CDLL(build_mock(), mode=C.RTLD_GLOBAL)   # takes the place of libhavoc_mi355x.so for everything loaded after it
    planes, stride = st.clip_planes(W, H, args.seed + 4, args.bit_depth)
    planes = [aligned(p) for p in planes]
    pad = 96
    pus = st.make_searches(W, H, args.searches, args.seed)
    par = st.medium_params(W, H, args.bit_depth)
    report = {"device": args.device, "res": args.res, "bit_depth": args.bit_depth, "searches": int(len(pus))}

    def by_list(fn):
        out = np.zeros(len(pus), st.RESULT_DT)
        for lst in (0, 1):
            sel = np.flatnonzero(pus["ref_list"] == lst)
            if len(sel):
                out[sel] = fn(np.ascontiguousarray(pus[sel]), lst)
        return out

    try:
        ref = st.Client("ref", 3)            # the reference's own library behind the table API
        report["expected_from"] = "reference tables (oracle/_ref)"
    except (FileNotFoundError, OSError):
        ref = st.Client("oracle")            # no oracle/_ref on this machine: the CPU oracle answers the per-block questions
        report["expected_from"] = "CPU oracle"
    t0 = time.perf_counter()
    expected = by_list(lambda sub, lst: ref.uni(par, planes[0], planes[1 + lst], stride, pad, sub))
    report["expected"] = {"seconds": round(time.perf_counter() - t0, 4), "calls": int(expected["calls"].sum()),
                          "calls_per_search": round(float(expected["calls"].mean()), 2)}
    # bi searches: the first `--bi` PUs, refined list = their ref_list, other list's vector and start vector from the uni results
    nbi = min(args.bi, len(pus))
    bi_pus = np.ascontiguousarray(pus[:nbi])
    start = np.ascontiguousarray(expected["mv"][:nbi])

    def run_bi(client):
        out = np.zeros(nbi, st.RESULT_DT)
        for lst in (0, 1):
            sel = np.flatnonzero(bi_pus["ref_list"] == lst)
            if len(sel):
                out[sel] = client.bi(par, planes[0], planes[1 + lst], planes[2 - lst], stride, pad, np.ascontiguousarray(bi_pus[sel]), start[sel])
        return out

    expected_bi = run_bi(ref)
    skip = set(args.skip.split(","))

    if "classic" not in skip:
        cl = st.Client("classic")
        # unregistered first: a few searches through the one-job launch path (the compatibility floor)
        few = np.ascontiguousarray(pus[:6])
        t0 = time.perf_counter()
        got_few = np.zeros(len(few), st.RESULT_DT)
        for lst in (0, 1):
            sel = np.flatnonzero(few["ref_list"] == lst)
            if len(sel):
                got_few[sel] = cl.uni(par, planes[0], planes[1 + lst], stride, pad, np.ascontiguousarray(few[sel]))
        t_few = time.perf_counter() - t0
        s0 = cl.stats()
        assert cl.register(planes[0], stride, pad, W, H, args.bit_depth, 0) == 0
        t0 = time.perf_counter()
        assert cl.register(planes[1], stride, pad, W, H, args.bit_depth, 1) == 0
        assert cl.register(planes[2], stride, pad, W, H, args.bit_depth, 1) == 0
        t_reg = time.perf_counter() - t0
        s1 = cl.stats()
        t0 = time.perf_counter()
        got = by_list(lambda sub, lst: cl.uni(par, planes[0], planes[1 + lst], stride, pad, sub))
        t_uni = time.perf_counter() - t0
        s2 = cl.stats()
        got_bi = run_bi(cl)
        s3 = cl.stats()
        # the reference calls its tables from one thread per CTU row (wavefront parallel processing): the same searches again,
        # `--threads` host threads each walking a slice of the list through the SAME table set at the same time
        from concurrent.futures import ThreadPoolExecutor

        def threaded(sub, lst):
            n, k = len(sub), max(1, args.threads)
            cuts = [n * t // k for t in range(k + 1)]
            out = np.zeros(n, st.RESULT_DT)
            with ThreadPoolExecutor(k) as pool:
                parts = list(pool.map(lambda t: cl.uni(par, planes[0], planes[1 + lst], stride, pad, sub, cuts[t], cuts[t + 1]), range(k)))
            for t in range(k):
                out[cuts[t]:cuts[t + 1]] = parts[t][cuts[t]:cuts[t + 1]]
            return out
        t0 = time.perf_counter()
        got_mt = by_list(threaded)
        t_mt = time.perf_counter() - t0
        s4 = cl.stats()
        calls = int(expected["calls"].sum())
        # every SAD / SAD4 call is one table call; every interpolate + SATD call is 1 + tiles table calls
        table_calls = (s2[0] - s1[0]) + (s2[1] - s1[1])
        report["classic"] = {
            "mismatching_searches": same(got, expected), "mismatching_bi": same(got_bi, expected_bi, ["mv", "mvd", "mvp_flag", "calls", "cost_subpel"]),
            "mismatching_unregistered": same(got_few, expected[:len(few)]),
            "unregistered": {"searches": int(len(few)), "seconds": round(t_few, 4), "launches": s0[2], "table_calls": s0[0] + s0[1]},
            "register_two_references_seconds": round(t_reg, 4), "bytes_uploaded": s1[6], "bytes_mirrored": s1[7],
            "uni": {"seconds": round(t_uni, 4), "table_calls": table_calls, "served": s2[0] - s1[0], "one_job_path": s2[1] - s1[1],
                    "launches": s2[2] - s1[2], "surfaces": s2[3] - s1[3], "satd_batches": s2[4] - s1[4],
                    "launches_per_search": round((s2[2] - s1[2]) / len(pus), 3), "us_per_table_call": round(t_uni / max(1, table_calls) * 1e6, 3),
                    "us_per_loop_call": round(t_uni / max(1, calls) * 1e6, 3)},
            "bi": {"searches": nbi, "served": s3[0] - s2[0], "one_job_path": s3[1] - s2[1], "launches": s3[2] - s2[2]},
            "threaded": {"threads": args.threads, "mismatching_searches": same(got_mt, expected), "seconds": round(t_mt, 4),
                         "served": s4[0] - s3[0], "one_job_path": s4[1] - s3[1], "launches": s4[2] - s3[2]},
        }

    if "batch" not in skip:
        dev = C.CDLL(os.path.join(st.BUILD, "mock", "libhavoc_mi355x.so") if args.device == "mock" else os.path.join(ROOT, "turingcodec_amd", "libhavoc_mi355x.so"),
                     mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(ROOT, "turingcodec_amd", "libhavoc_search.so"))
        vp, ip, i64 = C.c_void_p, C.c_ssize_t, C.c_int64
        dev.havoc_mi355x_create.argtypes = [C.POINTER(vp), C.c_int, vp]
        dev.havoc_mi355x_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
        dev.havoc_mi355x_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        dev.havoc_mi355x_interp_planes.argtypes = [vp, C.c_int, C.c_int, vp, ip, vp, ip, C.c_int, C.c_int, C.c_int, C.c_int]
        dev.havoc_mi355x_sync.argtypes = [vp]
        dev.havoc_mi355x_last_error.restype = C.c_char_p
        L.havoc_search_motion_uni.argtypes = [vp, C.c_int, C.POINTER(st.Params), vp, i64, ip, vp, i64, ip, C.c_int, vp, ip, i64, vp, C.c_int, vp, C.c_int,
                                              C.POINTER(Stats)]
        ctx = vp()
        rc = dev.havoc_mi355x_create(C.byref(ctx), 0, vp(-1 & 0xFFFFFFFFFFFFFFFF))
        assert rc == 0, dev.havoc_mi355x_last_error()
        n = planes[0].size
        pe = (n + 63) & ~63
        dplane = []
        for p in planes:
            d = vp()
            assert dev.havoc_mi355x_malloc(ctx, C.byref(d), n * S + 256) == 0
            assert dev.havoc_mi355x_h2d(ctx, d, p.ctypes.data, n * S) == 0
            dplane.append(d)
        dphase = []
        t0 = time.perf_counter()
        for r in (1, 2):
            d = vp()
            assert dev.havoc_mi355x_malloc(ctx, C.byref(d), 16 * pe * S + 256) == 0
            assert dev.havoc_mi355x_h2d(ctx, d, planes[r].ctypes.data, n * S) == 0
            assert dev.havoc_mi355x_interp_planes(ctx, S, args.bit_depth, d, pe, dplane[r], stride, 12, 4, W + 2 * pad - 24, H + 2 * pad - 8) == 0
            dphase.append(d)
        dev.havoc_mi355x_sync(ctx)
        t_planes = time.perf_counter() - t0
        origin = pad * stride + pad
        # twice: the first pass also allocates the client's pinned work memory (kept per context), the second is the steady state
        for attempt in (0, 1):
            stats = [Stats(), Stats()]
            got = np.zeros(len(pus), st.RESULT_DT)
            t0 = time.perf_counter()
            for lst in (0, 1):
                sel = np.flatnonzero(pus["ref_list"] == lst)
                sub = np.ascontiguousarray(pus[sel])
                out = np.zeros(len(sub), st.RESULT_DT)
                rc = L.havoc_search_motion_uni(ctx, S, C.byref(par), dplane[0], origin, stride, dplane[1 + lst], origin, stride, pad, dphase[lst], pe, origin,
                                               sub.ctypes.data, len(sub), out.ctypes.data, args.threads, C.byref(stats[lst]))
                assert rc == 0, (rc, dev.havoc_mi355x_last_error())
                got[sel] = out
            if attempt == 0:
                t_first = time.perf_counter() - t0
        t_batch = time.perf_counter() - t0
        # the same searches with the loops inside the kernel: ONE launch per list, a workgroup per search (the real device only)
        device = None
        if args.device == "real":
            L.havoc_search_motion_uni_device.argtypes = [vp, C.c_int, C.POINTER(st.Params), vp, i64, ip, vp, i64, ip, C.c_int, vp, ip, i64, vp, C.c_int, vp,
                                                         C.POINTER(Stats)]
            for attempt in (0, 1):
                dstats = [Stats(), Stats()]
                got_dev = np.zeros(len(pus), st.RESULT_DT)
                t0 = time.perf_counter()
                for lst in (0, 1):
                    sel = np.flatnonzero(pus["ref_list"] == lst)
                    sub = np.ascontiguousarray(pus[sel])
                    out = np.zeros(len(sub), st.RESULT_DT)
                    rc = L.havoc_search_motion_uni_device(ctx, S, C.byref(par), dplane[0], origin, stride, dplane[1 + lst], origin, stride, pad, dphase[lst], pe, origin,
                                                          sub.ctypes.data, len(sub), out.ctypes.data, C.byref(dstats[lst]))
                    assert rc == 0, (rc, dev.havoc_mi355x_last_error())
                    got_dev[sel] = out
                t_dev = time.perf_counter() - t0
            device = {"mismatching_searches": same(got_dev, expected), "seconds": round(t_dev, 5), "searches_per_second": round(len(pus) / t_dev, 1),
                      "launches": sum(s_.launches for s_ in dstats), "bytes_down": sum(s_.bytes_down for s_ in dstats)}
        # bi-directional refinement of the first `--bi` searches through the batch client: ideal predictors built on the device
        L.havoc_search_motion_bi.argtypes = [vp, C.c_int, C.POINTER(st.Params), vp, i64, ip, vp, i64, ip, C.c_int, vp, ip, i64, vp, i64, vp, vp, C.c_int, vp,
                                             C.c_int, C.POINTER(Stats)]
        got_bi_batch = np.zeros(nbi, st.RESULT_DT)
        bi_stats = [Stats(), Stats()]
        t0 = time.perf_counter()
        for lst in (0, 1):
            sel = np.flatnonzero(bi_pus["ref_list"] == lst)
            if not len(sel):
                continue
            sub = np.ascontiguousarray(bi_pus[sel])
            stt = np.ascontiguousarray(start[sel], np.int16)
            out = np.zeros(len(sub), st.RESULT_DT)
            rc = L.havoc_search_motion_bi(ctx, S, C.byref(par), dplane[0], origin, stride, dplane[1 + lst], origin, stride, pad, dphase[lst], pe, origin,
                                          dplane[2 - lst], origin, sub.ctypes.data, stt.ctypes.data, len(sub), out.ctypes.data, args.threads,
                                          C.byref(bi_stats[lst]))
            assert rc == 0, (rc, dev.havoc_mi355x_last_error())
            got_bi_batch[sel] = out
        t_bi = time.perf_counter() - t0
        tot = lambda f: sum(getattr(s_, f) for s_ in stats)
        report["batch"] = {
            "mismatching_searches": same(got, expected), "seconds": round(t_batch, 4), "seconds_first_call": round(t_first, 4),
            "phase_planes_seconds": round(t_planes, 4),
            "rounds": max(s_.rounds for s_ in stats), "launches": tot("launches"), "surfaces_small": tot("surfaces_small"),
            "surfaces_large": tot("surfaces_large"), "satd_jobs": tot("satd_jobs"), "replays": tot("replays"), "bytes_down": tot("bytes_down"),
            "seconds_gpu": round(tot("seconds_gpu"), 4), "seconds_host": round(tot("seconds_host"), 4),
            "searches_per_second": round(len(pus) / t_batch, 1), "loop_calls_per_second": round(int(expected["calls"].sum()) / t_batch, 1),
            "launches_per_search": round(tot("launches") / len(pus), 4), "threads": args.threads,
            "max_replays_of_one_search": int(got["replays"].max()),
            "loops_inside_the_kernel": device,
            "bi": {"searches": nbi, "mismatching": same(got_bi_batch, expected_bi, ["mv", "mvd", "mvp_flag", "calls", "cost_subpel"]), "seconds": round(t_bi, 4),
                   "launches": sum(s_.launches for s_ in bi_stats), "rounds": max(s_.rounds for s_ in bi_stats)},
        }
    # ---- 35-mode intra stage (Search.hpp:40-190): per-call through the reference tables vs one fused launch + host ordering
    if "intra" not in skip:
        from turingcodec_amd.workload import FrameWorkload
        wl = FrameWorkload(W, H, args.bit_depth, args.seed + 9)
        rsl = st.reciprocal_sqrt_lambda(32)
        src = aligned(wl.luma)
        intra = {"partitions": 0, "mismatching": [], "seconds_batch": 0.0, "seconds_reference": 0.0}
        if "batch" not in skip:
            vp, ip = C.c_void_p, C.c_ssize_t
            L.havoc_search_intra_modes.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, ip, vp, vp, C.c_int, vp, C.c_double, vp, vp]
            dsrc = vp()
            assert dev.havoc_mi355x_malloc(ctx, C.byref(dsrc), src.nbytes + 256) == 0
            assert dev.havoc_mi355x_h2d(ctx, dsrc, src.ctypes.data, src.nbytes) == 0
        for log2 in (2, 3, 4, 5):
            jobs = np.ascontiguousarray(wl.intra_search[log2][:160 if W < 1000 else 4000])
            nb = aligned(wl.intra_search_nb[log2])
            n = len(jobs)
            if not n:
                continue
            ictx = st.make_intra_contexts(n, log2, args.seed + log2)
            t0 = time.perf_counter()
            satd_ref = ref.intra35(args.bit_depth, log2, src, wl.stride, nb, jobs)
            exp = ref.intra_order(ictx, rsl, satd_ref)
            intra["seconds_reference"] += time.perf_counter() - t0
            intra["partitions"] += n
            if "batch" not in skip:
                dnb, djobs = vp(), vp()
                assert dev.havoc_mi355x_malloc(ctx, C.byref(dnb), nb.nbytes + 256) == 0 and dev.havoc_mi355x_malloc(ctx, C.byref(djobs), jobs.nbytes + 256) == 0
                assert dev.havoc_mi355x_h2d(ctx, dnb, nb.ctypes.data, nb.nbytes) == 0 and dev.havoc_mi355x_h2d(ctx, djobs, jobs.ctypes.data, jobs.nbytes) == 0
                out = np.zeros(n, st.INTRA_RESULT_DT)
                satd = np.zeros((n, 35), np.int32)
                t0 = time.perf_counter()
                rc = L.havoc_search_intra_modes(ctx, S, args.bit_depth, log2, dsrc, wl.stride, dnb, djobs, n, ictx.ctypes.data, rsl, out.ctypes.data, satd.ctypes.data)
                intra["seconds_batch"] += time.perf_counter() - t0
                assert rc == 0, rc
                if not (np.array_equal(satd, satd_ref) and out.tobytes() == exp.tobytes()):
                    intra["mismatching"].append(log2)
            if "classic" not in skip and log2 == 3:
                few = 4      # the table API has no batched intra entry: 35 x (1 + tiles) one-job launches per partition
                got = cl.intra35(args.bit_depth, log2, src, wl.stride, nb, jobs[:few])
                if not np.array_equal(got, satd_ref[:few]):
                    intra["mismatching"].append("classic")
        intra["seconds_batch"] = round(intra["seconds_batch"], 4)
        intra["seconds_reference"] = round(intra["seconds_reference"], 4)
        intra["mode_decisions_per_second_batch"] = round(intra["partitions"] / max(1e-9, intra["seconds_batch"]), 1) if "batch" not in skip else None
        report["intra"] = intra
    print(json.dumps(report))


if __name__ == "__main__":
    main()
