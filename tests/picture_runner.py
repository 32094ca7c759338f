"""Runs the PICTURE client of libhavoc_search.so (a whole picture's motion searches in wavefront order with predictors derived from
earlier decisions, turingcodec_amd/search/picture_search.cpp) against the same walk made one table call at a time over the reference's
tables, in a subprocess of the tests (the device library is chosen at load time):

  --device mock : tests/mock_device.c stands in for libhavoc_mi355x.so (host logic, no GPU)
  --device real : the MI355X library

Prints one JSON line: mismatching searches (every field of every (PU, list) result), whether the final motion fields agree, the
client's statistics (wavefront steps, rounds, launches, bytes, seconds) of the second (steady-state) call.
"""
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
from search_runner import aligned, build_mock, same  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", choices=["mock", "real"], default="real")
    ap.add_argument("--res", default="416x240")
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--density", type=float, default=1.0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--expected", choices=["ref", "oracle", "none"], default="ref")
    ap.add_argument("--ref-mask", type=int, default=3, help="instruction sets of the reference's tables the expected walk runs through: 3 = its plain-C functions, "
                    "-1 = everything the host supports (the x86 JIT code: same results, several times faster -- for the 4K / 8K pictures)")
    ap.add_argument("--distance", type=int, default=1, help="temporal distance of the two reference pictures")
    ap.add_argument("--speed", choices=["slow", "medium", "fast"], default="medium", help="turing/Speed.h: medium = early termination (MET), +-64 window, half + "
                    "quarter refinement; fast = also the small windows and no quarter-sample step; slow = no early termination (every search runs the star "
                    "and, where it travels, the raster refinement)")
    ap.add_argument("--stress", type=int, default=0, help="run the device search (with the bi-directional refinement) this many more times and count the "
                    "runs whose results differ from the expected ones (workgroups of a picture wait for each other: a race would show as a run that differs)")
    args = ap.parse_args()
    W, H = (int(v) for v in args.res.split("x"))
    S = 1 if args.bit_depth == 8 else 2
    if args.device == "mock":
        C.CDLL(build_mock(), mode=C.RTLD_GLOBAL)   # takes the place of libhavoc_mi355x.so for everything loaded after it
    from turingcodec_amd import decisions, workload
    planes, stride = st.clip_planes(W, H, args.seed + 4, args.bit_depth, distance=args.distance)
    planes = [aligned(p) for p in planes]
    pad = 96
    pus, first, cx, cy = workload.picture_pus(W, H, args.seed, args.density)
    par = st.medium_params(W, H, args.bit_depth, args.qp)
    if args.speed == "fast":
        par.small_search_window, par.bi_small_search_window, par.quarter_pel = 1, 1, 0
    elif args.speed == "slow":
        par.met = 0
    rate = (45000, 98000)      # rate of mvp_lX_flag = 0 / 1 in some CABAC state (Q16 bits)
    report = {"device": args.device, "res": args.res, "bit_depth": args.bit_depth, "speed": args.speed, "distance": args.distance, "pus": int(len(pus)), "searches": int(2 * len(pus)), "ctus": cx * cy}

    expected = None
    if args.expected != "none":
        try:
            ref = st.Client("ref", args.ref_mask) if args.expected == "ref" else st.Client("oracle")
            report["expected_from"] = "reference tables (oracle/_ref)" if args.expected == "ref" else "CPU oracle"
        except (FileNotFoundError, OSError):
            ref = st.Client("oracle")
            report["expected_from"] = "CPU oracle"
        t0 = time.perf_counter()
        expected, expected_field, expected_bi = ref.picture_uni(par, planes[0], planes[1], planes[2], stride, pad, pus, first, cx, cy, rate, bi=True)
        report["bi_loop_calls"] = int(expected_bi["calls"].sum())
        report["expected_seconds"] = round(time.perf_counter() - t0, 4)
        report["loop_calls"] = int(expected["calls"].sum())

    dev = C.CDLL(os.path.join(st.BUILD, "mock", "libhavoc_mi355x.so") if args.device == "mock" else os.path.join(ROOT, "turingcodec_amd", "libhavoc_mi355x.so"),
                 mode=C.RTLD_GLOBAL)
    vp, ip = C.c_void_p, C.c_ssize_t
    dev.havoc_mi355x_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    dev.havoc_mi355x_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    dev.havoc_mi355x_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    dev.havoc_mi355x_interp_planes.argtypes = [vp, C.c_int, C.c_int, vp, ip, vp, ip, C.c_int, C.c_int, C.c_int, C.c_int]
    dev.havoc_mi355x_sync.argtypes = [vp]
    dev.havoc_mi355x_last_error.restype = C.c_char_p
    ctx = vp()
    rc = dev.havoc_mi355x_create(C.byref(ctx), 0, vp(-1 & 0xFFFFFFFFFFFFFFFF))
    assert rc == 0, dev.havoc_mi355x_last_error()
    n = planes[0].size
    pe = (n + 63) & ~63
    # one allocation for the three pictures (source, list 0, list 1), one for the 2 x 16 phase planes: a job names a plane by a 32-bit offset
    dpic = vp()
    assert dev.havoc_mi355x_malloc(ctx, C.byref(dpic), 3 * pe * S + 256) == 0
    for k, p in enumerate(planes):
        assert dev.havoc_mi355x_h2d(ctx, dpic.value + k * pe * S, p.ctypes.data, n * S) == 0
    dphase = vp()
    assert dev.havoc_mi355x_malloc(ctx, C.byref(dphase), 32 * pe * S + 256) == 0
    t0 = time.perf_counter()
    for r in (0, 1):
        base = dphase.value + r * 16 * pe * S
        assert dev.havoc_mi355x_h2d(ctx, base, planes[1 + r].ctypes.data, n * S) == 0
        assert dev.havoc_mi355x_interp_planes(ctx, S, args.bit_depth, base, pe, dpic.value + (1 + r) * pe * S, stride, 12, 4, W + 2 * pad - 24, H + 2 * pad - 8) == 0
    dev.havoc_mi355x_sync(ctx)
    report["phase_planes_seconds"] = round(time.perf_counter() - t0, 4)
    origin = pad * stride + pad
    for attempt in range(args.repeat):
        t0 = time.perf_counter()
        got, field, stats = decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, pad, dphase.value, pe,
                                                  (origin, 16 * pe + origin), pus, first, cx, cy, rate, args.threads)
        t = time.perf_counter() - t0
        if attempt == 0:
            report["seconds_first_call"] = round(t, 4)
    d = stats.as_dict()
    d.update({"seconds": round(t, 5), "pictures_per_second": round(1.0 / t, 2), "searches_per_second": round(2 * len(pus) / t, 1),
              "rounds_per_step": round(d["rounds"] / max(1, d["steps"]), 2), "launches_per_step": round(d["launches"] / max(1, d["steps"]), 2),
              "max_replays_of_one_search": int(got["replays"].max()), "threads": args.threads})
    for k in ("seconds_gpu", "seconds_host", "seconds_total"):
        d[k] = round(d[k], 5)
    report["picture"] = d
    if args.device == "mock":
        # the device-resident search is a kernel: the mock has nothing behind it.  What can be checked without a GPU is the host wrapper in front
        # of it -- validation of the PU list, staging, the call -- ending in the stub's refusal; and the wrapper's own refusals
        try:
            decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, pad, dphase.value, pe,
                                  (origin, 16 * pe + origin), pus, first, cx, cy, rate, on_device=True, bi=True)
            report["device_wrapper_on_mock"] = "returned"
        except RuntimeError as e:
            report["device_wrapper_on_mock"] = str(e)
        refused = []
        bad = pus.copy()
        bad["x0"][0] = W      # outside the picture
        for label, kw, p_ in (("pu_outside", {}, bad), ("short_border", {"ref_pad": 64}, pus)):
            try:
                decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, kw.get("ref_pad", pad), dphase.value, pe,
                                      (origin, 16 * pe + origin), p_, first, cx, cy, rate, on_device=True)
            except RuntimeError as e:
                refused.append(label + ": " + str(e))
        report["device_wrapper_refusals"] = refused
    if args.device == "real":
        # the same picture with the decision loops inside the kernel (havoc_search_picture_uni_device): no rounds, one launch per wavefront step
        for form in ("steps", "rows"):      # one launch per wavefront step / one launch, rows waiting for each other inside the kernel (the default)
            os.environ["HAVOC_SEARCH_STEP_LAUNCHES"] = "1" if form == "steps" else "0"
            for attempt in range(args.repeat):
                t0 = time.perf_counter()
                got_d, field_d, stats_d = decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, pad,
                                                                dphase.value, pe, (origin, 16 * pe + origin), pus, first, cx, cy, rate, on_device=True)
                t = time.perf_counter() - t0
            if form == "steps":
                report["on_device_step_launches"] = {"seconds": round(t, 5), "launches": int(stats_d.launches),
                                                     "mismatches_vs_batch_client": len(same(got_d, got)), "field_equal_batch_client": bool(np.array_equal(field_d, field))}
        report["on_device"] = {"seconds": round(t, 5), "pictures_per_second": round(1.0 / t, 2), "searches_per_second": round(2 * len(pus) / t, 1),
                               "launches": int(stats_d.launches), "bytes_down": int(stats_d.bytes_down)}
        # ... and with the bi-directional refinement of every PU after its two searches (searchBi), the two lists' workgroups meeting per PU
        for attempt in range(args.repeat):
            t0 = time.perf_counter()
            got_b, field_b, stats_b, bi_b = decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, pad,
                                                                  dphase.value, pe, (origin, 16 * pe + origin), pus, first, cx, cy, rate, on_device=True, bi=True)
            t = time.perf_counter() - t0
        report["on_device_with_bi"] = {"seconds": round(t, 5), "pictures_per_second": round(1.0 / t, 2), "refinements": int((bi_b["calls"] > 0).sum()),
                                       "uni_mismatches_vs_without_bi": len(same(got_b, got_d)), "field_equal": bool(np.array_equal(field_b, field_d))}
        if expected is not None:
            bi_fields = ["mv", "mvd", "mvp_flag", "calls", "cost_subpel"]
            report["on_device_with_bi"]["mismatches"] = len(same(bi_b, expected_bi, bi_fields))
            report["on_device_with_bi"]["mismatching"] = same(bi_b, expected_bi, bi_fields)[:10]
        if args.stress and expected is not None:
            bad_runs = 0
            for _ in range(args.stress):
                g2, f2, _, b2 = decisions.picture_uni(ctx, S, par, dpic.value, origin, stride, dpic.value, (pe + origin, 2 * pe + origin), stride, pad, dphase.value, pe,
                                                      (origin, 16 * pe + origin), pus, first, cx, cy, rate, on_device=True, bi=True)
                bad_runs += int(bool(same(g2, expected)) or bool(same(b2, expected_bi, ["mv", "mvd", "mvp_flag", "calls", "cost_subpel"]))
                                or not np.array_equal(f2, expected_field))
            report["stress"] = {"runs": args.stress, "runs_that_differ": bad_runs}
        fields = [k for k in got.dtype.names if k != "replays"]
        report["on_device"]["mismatches_vs_batch_client"] = int(sum(any(not np.array_equal(got_d[k][i], got[k][i]) for k in fields) for i in range(len(got))))
        report["on_device"]["field_equal_batch_client"] = bool(np.array_equal(field_d, field))
        if expected is not None:
            report["on_device"]["mismatches"] = len(same(got_d, expected))
            report["on_device"]["mismatching_searches"] = same(got_d, expected)[:10]
            report["on_device"]["field_equal"] = bool(np.array_equal(field_d, expected_field))
    if expected is not None:
        report["mismatching_searches"] = same(got, expected)[:20]
        report["mismatches"] = len(same(got, expected))
        report["field_equal"] = bool(np.array_equal(field, expected_field))
    print(json.dumps(report))


if __name__ == "__main__":
    main()
