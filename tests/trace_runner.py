#!/usr/bin/env python3
"""Replays the reference encoder's own motion searches, bi-directional refinements and intra mode orders (tests/trace_tools.py: the traced
reference encoder oracle/_ref/turing_ref_trace) through turingcodec_amd/search/decision.hpp in ONE fresh process and prints a JSON report
(tests/test_trace_pin.py asserts on it; profiles/ keeps the GPU box's report).

  cpu            decision.hpp per call over the reference's havoc tables (tests/search_client.cpp, logged): the CALL SEQUENCE (every sad / sad4 /
                 interpolate + satd position and value) and every decision must be the reference's
  --device mock  + the batch clients of libhavoc_search.so (launch + host replay) over the CPU stand-in device (tests/mock_device.c)
  --device real  + the same on the MI355X, and the searches with the loops INSIDE the kernels: havoc_search_motion_uni_device (k_search_list),
                 havoc_search_motion_bi_device (k_search_bi_list), havoc_mi355x_intra_order (k_intra_order)
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import encoder_tools as et  # noqa: E402
import search_tools as st  # noqa: E402
import trace_tools as tt  # noqa: E402

UNI_FIELDS = ["mv", "mvd", "mv_integer", "mvp_flag", "cost_integer", "calls"]
BI_FIELDS = ["mv", "mvd", "mvp_flag", "cost_subpel", "calls"]


def differing(got, exp, fields):
    if not len(got):
        return np.zeros(0, np.int64)
    return np.flatnonzero(~np.all([np.all(got[f].reshape(len(got), -1) == exp[f].reshape(len(exp), -1), axis=1) for f in fields], axis=0))


def gather_rows(trace, sel):
    if not len(sel):
        return np.zeros((0, 13), np.int32)
    return np.concatenate([trace.rows[trace.first[i]:trace.first[i + 1]] for i in sel])


class Stats(C.Structure):
    _fields_ = [("rounds", C.c_int32), ("launches", C.c_int32), ("surfaces_small", C.c_int32), ("surfaces_large", C.c_int32),
                ("satd_jobs", C.c_int32), ("replays", C.c_int32), ("bytes_down", C.c_int64), ("seconds_gpu", C.c_double),
                ("seconds_host", C.c_double), ("seconds_total", C.c_double)]


class Device:
    """libhavoc_mi355x.so (or the stand-in) + libhavoc_search.so, planes uploaded once per picture"""

    def __init__(self, kind):
        import search_runner
        self.kind = kind
        path = search_runner.build_mock() if kind == "mock" else os.path.join(ROOT, "turingcodec_amd", "libhavoc_mi355x.so")
        dev = self.dev = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L = self.L = C.CDLL(os.path.join(ROOT, "turingcodec_amd", "libhavoc_search.so"))
        vp, ip, i64, i = C.c_void_p, C.c_ssize_t, C.c_int64, C.c_int
        dev.havoc_mi355x_create.argtypes = [C.POINTER(vp), i, vp]
        dev.havoc_mi355x_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
        dev.havoc_mi355x_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        dev.havoc_mi355x_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        dev.havoc_mi355x_interp_planes.argtypes = [vp, i, i, vp, ip, vp, ip, i, i, i, i]
        dev.havoc_mi355x_sync.argtypes = [vp]
        dev.havoc_mi355x_last_error.restype = C.c_char_p
        dev.havoc_mi355x_intra_order.argtypes = [vp, vp, vp, i, C.c_int32, vp, vp, vp, vp]
        P = C.POINTER(st.Params)
        L.havoc_search_motion_uni.argtypes = [vp, i, P, vp, i64, ip, vp, i64, ip, i, vp, ip, i64, vp, i, vp, i, C.POINTER(Stats)]
        L.havoc_search_motion_uni_device.argtypes = [vp, i, P, vp, i64, ip, vp, i64, ip, i, vp, ip, i64, vp, i, vp, C.POINTER(Stats)]
        L.havoc_search_motion_bi.argtypes = [vp, i, P, vp, i64, ip, vp, i64, ip, i, vp, ip, i64, vp, i64, vp, vp, i, vp, i, C.POINTER(Stats)]
        L.havoc_search_motion_bi_device.argtypes = [vp, i, P, vp, i64, ip, vp, i64, ip, i, vp, ip, i64, vp, i64, vp, vp, i, vp, C.POINTER(Stats)]
        self.ctx = vp()
        rc = dev.havoc_mi355x_create(C.byref(self.ctx), 0, vp(-1 & 0xFFFFFFFFFFFFFFFF))
        assert rc == 0, dev.havoc_mi355x_last_error()
        self.planes = {}     # (what, poc) -> (device pointer, host array)

    def _alloc(self, nbytes):
        d = C.c_void_p()
        assert self.dev.havoc_mi355x_malloc(self.ctx, C.byref(d), nbytes + 256) == 0
        return d

    def plane(self, what, poc, host):
        key = (what, poc)
        if key not in self.planes:
            d = self._alloc(host.nbytes)
            assert self.dev.havoc_mi355x_h2d(self.ctx, d, host.ctypes.data, host.nbytes) == 0
            self.planes[key] = d
        return self.planes[key]

    def phase(self, poc, host, stride, width, height, bit_depth):
        """the 16 fractional-sample planes of a reference picture (plane 0 = the picture): (device pointer, plane_elems)"""
        key = ("phase", poc)
        S = host.itemsize
        pe = (host.size + 63) & ~63
        if key not in self.planes:
            d = self._alloc(16 * pe * S)
            assert self.dev.havoc_mi355x_h2d(self.ctx, d, host.ctypes.data, host.nbytes) == 0
            assert self.dev.havoc_mi355x_interp_planes(self.ctx, S, bit_depth, d, pe, self.plane("ref", poc, host), stride, 12, 4, width + 2 * tt.PAD - 24,
                                                       height + 2 * tt.PAD - 8) == 0
            self.dev.havoc_mi355x_sync(self.ctx)
            self.planes[key] = d
        return self.planes[key], pe


def z_order(x4, y4):
    """Morton index of a 4x4 cell inside its 64x64 CTU"""
    z = 0
    for b in range(4):
        z |= ((x4 >> b) & 1) << (2 * b) | ((y4 >> b) & 1) << (2 * b + 1)
    return z


def intra_neighbours(intra, w, h, bit_depth, dev):
    """the reference samples the ENCODER held for every intra partition it searched (trace records INTRA_NB / INTRA_NBF: Search.hpp:57-59) against
      * IntraReferenceSamples::filter restated (oracle_intra_filter_neighbours; strong intra smoothing = the encoder's default, Encoder.cpp:688): filtered == F(unfiltered);
      * the substitution process (HEVC 8.4.4.2.2) on the availability of HEVC 6.4.1 (inside the picture, earlier in z-scan order): the recorded array rebuilt from its
        available samples alone must be the recorded array (a sample the encoder had substituted although it is available by 6.4.1 cannot be seen this way, the
        other direction can);
      * with a device: havoc_mi355x_intra_gather (k_intra_gather / the stand-in) on a picture tiled with the recorded samples: its unfiltered and filtered arrays."""
    from reflibs import Oracle
    orc = Oracle()
    sel = [i for i in range(len(intra)) if intra.neighbours[i] is not None]
    r = {"partitions_with_samples": len(sel), "filtered_arrays": 0, "filter_mismatching": 0, "strong_smoothing_taken": 0, "substitution_mismatching": 0,
         "partitions_with_substituted_samples": 0}
    ctus_x = (w + 63) // 64
    for i in sel:
        poc, x0, y0, log2 = (int(v) for v in intra.where[i])
        n = 1 << log2
        unf, fil = intra.neighbours[i], intra.neighbours_filtered[i]
        if fil is not None:
            r["filtered_arrays"] += 1
            want = orc.intra_filter_neighbours(unf, n, bit_depth, 1)
            r["filter_mismatching"] += int(not np.array_equal(want, fil))
            r["strong_smoothing_taken"] += int(n == 32 and not np.array_equal(want, orc.intra_filter_neighbours(unf, n, bit_depth, 0)))
        k = np.arange(4 * n + 1)
        x = np.where(k <= 2 * n, x0 - 1, x0 + k - 2 * n - 1)
        y = np.where(k < 2 * n, y0 + 2 * n - 1 - k, y0 - 1)
        inside = (x >= 0) & (y >= 0) & (x < w) & (y < h)
        addr = lambda xx, yy: ((yy >> 6) * ctus_x + (xx >> 6)) * 256 + z_order((xx >> 2) & 15, (yy >> 2) & 15)
        me = addr(x0, y0)
        have = np.array([bool(inside[j]) and addr(int(x[j]), int(y[j])) < me for j in range(len(k))])
        r["partitions_with_substituted_samples"] += int(not have.all())
        rebuilt = orc.intra_substitute(np.where(have, unf, 0), have, n, bit_depth)
        r["substitution_mismatching"] += int(not np.array_equal(rebuilt, unf))
    if dev is None or not sel:
        return r
    # ---- the device's gather on a picture made of the recorded samples: a 192 x 192 tile per partition, the partition at (64, 64) of its tile
    sel = [i for i in sel if intra.neighbours_filtered[i] is not None][:4000]
    S = 1 if bit_depth == 8 else 2
    dt = np.uint8 if S == 1 else np.uint16
    T = 192
    cols = 16
    rows = (len(sel) + cols - 1) // cols
    W, H = cols * T, rows * T
    PADG = 4                                                                          # (the C ABI wants a border of at least one sample)
    full = np.zeros((H + 2 * PADG, W + 2 * PADG), dt)
    pic = full[PADG:PADG + H, PADG:PADG + W]
    parts = np.zeros((len(sel), 4), np.int32)
    jobs = np.zeros((len(sel), 8), np.int32)
    for t, i in enumerate(sel):
        n = 1 << int(intra.where[i][3])
        ox, oy = (t % cols) * T + 64, (t // cols) * T + 64
        unf = intra.neighbours[i]
        pic[oy + 2 * n - 1 - np.arange(2 * n), ox - 1] = unf[:2 * n]
        pic[oy - 1, ox - 1] = unf[2 * n]
        pic[oy - 1, ox + np.arange(2 * n)] = unf[2 * n + 1:]
        parts[t] = (ox, oy, int(intra.where[i][3]), 1)
        jobs[t, 1], jobs[t, 2] = 264 * (2 * t) + 132, 264 * (2 * t + 1) + 132      # nb_off / nbf_off: the arrays' middles
    owner = np.zeros((H // 4, W // 4), np.int32)                                      # every cell precedes the partitions (index 1): all samples available
    modes = np.ones((H // 4, W // 4), np.uint8)
    layout = (C.c_int32 * 8)(W, H, W + 2 * PADG, PADG, W // 4, bit_depth, 6, 1)
    nb = np.zeros(264 * 2 * len(sel) + 264, dt)
    mpm = np.zeros(len(sel), st.INTRA_CTX_DT)
    d = {}
    for name, a in (("pic", full), ("owner", owner), ("modes", modes), ("parts", parts), ("jobs", jobs), ("nb", nb), ("mpm", mpm)):
        d[name] = dev._alloc(a.nbytes)
        assert dev.dev.havoc_mi355x_h2d(dev.ctx, d[name], a.ctypes.data, a.nbytes) == 0
    vp = C.c_void_p
    dev.dev.havoc_mi355x_intra_gather.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    rc = dev.dev.havoc_mi355x_intra_gather(dev.ctx, S, layout, d["pic"], d["owner"], d["modes"], d["parts"], len(sel), d["jobs"], d["nb"], d["mpm"])
    assert rc == 0, dev.dev.havoc_mi355x_last_error()
    dev.dev.havoc_mi355x_sync(dev.ctx)
    assert dev.dev.havoc_mi355x_d2h(dev.ctx, nb.ctypes.data, d["nb"], nb.nbytes) == 0
    r.update({"gathered": len(sel), "gather_unfiltered_mismatching": 0, "gather_filtered_mismatching": 0})
    for t, i in enumerate(sel):
        n = 1 << int(intra.where[i][3])
        lo = 2 * n + 1
        got_u = nb[264 * (2 * t) + 132 - lo:264 * (2 * t) + 132 - lo + 4 * n + 1].astype(np.int32)
        got_f = nb[264 * (2 * t + 1) + 132 - lo:264 * (2 * t + 1) + 132 - lo + 4 * n + 1].astype(np.int32)
        r["gather_unfiltered_mismatching"] += int(not np.array_equal(got_u, intra.neighbours[i]))
        r["gather_filtered_mismatching"] += int(not np.array_equal(got_f, intra.neighbours_filtered[i]))
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--device", choices=["none", "mock", "real"], default="none")
    ap.add_argument("--threads", type=int, default=1, help="encoder threads of the traced run")
    ap.add_argument("--limit", type=int, default=0, help="device legs: at most this many searches per kind (0: all)")
    args = ap.parse_args()
    w, h, nframes, seed, bd, opts = et.CASES[args.case]
    report = {"case": args.case, "device": args.device, "encoder_threads": args.threads}
    with tempfile.TemporaryDirectory() as work:
        t0 = time.perf_counter()
        stream, records, recon, internal = tt.run(args.case, work, args.threads)
        report["encode_seconds"] = round(time.perf_counter() - t0, 2)
    g = et.golden().get(args.case)
    report["stream_is_the_committed_reference_stream"] = bool(g and et.md5(stream) == g["stream_md5"] and len(stream) == g["stream_bytes"])
    report["trace_records"] = int(len(records))
    src = tt.source_frames(args.case, internal)
    uni, bi, intra = tt.MotionTrace(records), tt.MotionTrace(records, bi=True), tt.IntraTrace(records)
    rqt = tt.RqtTrace(records)
    amvp = tt.AmvpTrace(records)
    merge = tt.MergeTrace(records)
    del records
    ref_of = {}          # (poc, list) -> reference picture's poc, from the uni searches (a bi refinement follows the two uni searches of its PU)
    for poc, lst, rp in zip(uni.meta["poc"], uni.pus["ref_list"], uni.meta["ref_poc"]):
        ref_of[(int(poc), int(lst))] = int(rp)
    padded_src, padded_rec = {}, {}

    def P(cache, frames, poc):
        if poc not in cache:
            cache[poc] = tt.padded(frames[poc][0])
        return cache[poc]

    try:
        cpu = st.Client("ref", 3)
        report["cpu_tables"] = "the reference's havoc tables (oracle/_ref)"
    except (FileNotFoundError, OSError):
        cpu = st.Client("oracle")
        report["cpu_tables"] = "CPU oracle"

    # ---- CPU: decision.hpp per call, logged
    t0 = time.perf_counter()
    r = {"searches": int(len(uni)), "calls": int(len(uni.rows)), "mismatching_searches": 0, "mismatching_call_rows": 0, "groups": 0}
    expected_uni = {}
    for key, idx in uni.groups().items():
        par = uni.params(key, w, h)
        (s, stride), (rf, _) = P(padded_src, src, key[0]), P(padded_rec, recon, key[1])
        pus = np.ascontiguousarray(uni.pus[idx])
        out, rows, first = cpu.uni_logged(par, s, rf, stride, tt.PAD, pus, int(uni.results["calls"][idx].sum()) + 4096)
        r["mismatching_searches"] += int(len(differing(out, uni.results[idx], UNI_FIELDS)))
        r["mismatching_in_the_lane_formulation"] = r.get("mismatching_in_the_lane_formulation", 0) + int(len(differing(cpu.uni_lanes(par, s, rf, stride, tt.PAD, pus), uni.results[idx], UNI_FIELDS)))
        exp_rows = gather_rows(uni, idx)
        r["mismatching_call_rows"] += int(abs(len(rows) - len(exp_rows)) + np.count_nonzero(np.any(rows[:len(exp_rows)] != exp_rows[:len(rows)], axis=1)))
        r["groups"] += 1
        expected_uni[key] = (idx, out)
    # mvPreviousInteger2Nx2N: a search that reports wrote_2Nx2N must be what the NEXT search of the same CTU row and list starts from (Search.hpp:2170-2176, 2332-2335)
    chain_checked = chain_bad = 0
    state = {}
    order = np.argsort(uni.meta["seq"], kind="stable")      # the order the encoder ran them in, whatever thread a CTU ran on
    wrote = np.zeros(len(uni), np.int16)
    for key, (idx, out) in expected_uni.items():
        wrote[idx] = out["wrote_2Nx2N"]
    for i in order:
        k = (int(uni.meta["poc"][i]), int(uni.pus["y_ctb"][i]), int(uni.pus["ref_list"][i]))
        prev = tuple(int(v) for v in uni.pus["mv_previous_2Nx2N"][i])
        if k in state:
            chain_checked += 1
            chain_bad += prev != state[k]
        state[k] = tuple(int(v) for v in uni.results["mv_integer"][i]) if wrote[i] else prev
    r["previous_2Nx2N_handovers_checked"], r["previous_2Nx2N_handovers_wrong"] = chain_checked, int(chain_bad)
    r["seconds"] = round(time.perf_counter() - t0, 2)
    report["uni_cpu"] = r

    t0 = time.perf_counter()
    r = {"searches": int(len(bi)), "calls": int(len(bi.rows)), "mismatching_searches": 0, "mismatching_call_rows": 0}
    bi_jobs = []
    for key, idx in bi.groups().items():
        par = bi.params(key, w, h)
        for lst in (0, 1):
            sel = idx[bi.pus["ref_list"][idx] == lst]
            if not len(sel):
                continue
            other = ref_of[(key[0], 1 - lst)]
            (s, stride), (rf, _), (ro, _) = P(padded_src, src, key[0]), P(padded_rec, recon, key[1]), P(padded_rec, recon, other)
            pus = np.ascontiguousarray(bi.pus[sel])
            start = np.ascontiguousarray(bi.meta["start"][sel])
            out, rows, first = cpu.bi_logged(par, s, rf, ro, stride, tt.PAD, pus, start, int(bi.results["calls"][sel].sum()) + 4096)
            r["mismatching_searches"] += int(len(differing(out, bi.results[sel], BI_FIELDS)))
            r["mismatching_in_the_lane_formulation"] = r.get("mismatching_in_the_lane_formulation", 0) + int(len(differing(cpu.bi_lanes(par, s, rf, ro, stride, tt.PAD, pus, start), bi.results[sel], BI_FIELDS)))
            exp_rows = gather_rows(bi, sel)
            r["mismatching_call_rows"] += int(abs(len(rows) - len(exp_rows)) + np.count_nonzero(np.any(rows[:len(exp_rows)] != exp_rows[:len(rows)], axis=1)))
            bi_jobs.append((key, par, lst, other, sel, pus, start))
    r["seconds"] = round(time.perf_counter() - t0, 2)
    report["bi_cpu"] = r

    r = {"partitions": int(len(intra)), "mismatching": 0, "by_log2_size": {int(k): int(v) for k, v in zip(*np.unique(intra.where[:, 3], return_counts=True))} if len(intra) else {}}
    live = np.arange(35)[None, :]
    for rsl in np.unique(intra.rsl):
        sel = np.flatnonzero(intra.rsl == rsl)
        out = cpu.intra_order(np.ascontiguousarray(intra.ctx[sel]), float(rsl), intra.satd[sel])
        bad = (out["count"] != intra.count[sel]) | np.any((out["order"] != intra.order[sel]) & (live < intra.count[sel][:, None]), axis=1) | \
            np.any(out["costs"] != intra.costs[sel], axis=1)
        r["mismatching"] += int(bad.sum())
    report["intra_cpu"] = r

    report["intra_neighbours_cpu"] = intra_neighbours(intra, w, h, internal, None)

    # ---- amvp.hpp (the reference's two-predictor derivation restated as data-only code) on the encoder's own neighbours: the predictors it derived
    if len(amvp):
        got = cpu.amvp(amvp.rows)
        bad = np.flatnonzero(np.any(got != amvp.mvp, axis=1))
        nbr = amvp.rows[:, 4:49].reshape(-1, 5, 9)
        target = amvp.rows[:, 2][:, None]
        own = amvp.rows[:, 0]
        into_target = ((nbr[:, :, 1] != 0) & (nbr[:, :, 3] == target)) | ((nbr[:, :, 2] != 0) & (nbr[:, :, 4] == target))
        # how far picture_order.hpp's two-candidate stand-in (A1's and B1's vector of the same list, duplicate pruned, zero-filled) is from the real rule on the same neighbours
        X = amvp.rows[:, 0]
        pick = lambda k: (np.where(X == 0, nbr[:, k, 1], nbr[:, k, 2]) != 0) & (nbr[:, k, 0] != 0)
        vec = lambda k: np.where((X == 0)[:, None], nbr[:, k, 5:7], nbr[:, k, 7:9])
        ha, hb, va, vb = pick(1), pick(3), vec(1), vec(3)
        s0 = np.where(ha[:, None], va, np.where(hb[:, None], vb, 0))
        s1 = np.where((ha & hb & np.any(va != vb, axis=1))[:, None], vb, 0)
        stand_in = np.concatenate([s0, s1], axis=1)
        report["amvp_cpu"] = {"derivations": int(len(amvp)), "mismatching": int(len(bad)),
                              "two_candidate_stand_in_of_picture_order_hpp_agrees": int(np.sum(np.all(stand_in == amvp.mvp, axis=1))),
                              "with_a_scaled_candidate": int(np.sum(np.any((nbr[:, :, 0] != 0) & ~into_target, axis=1))),
                              "with_the_temporal_candidate_available": int((amvp.rows[:, 49] != 0).sum()),
                              "with_no_neighbour_at_all": int(np.sum(~np.any(nbr[:, :, 0] != 0, axis=1))),
                              "second_predictor_is_not_zero": int(np.sum(np.any(amvp.mvp[:, 2:] != 0, axis=1))),
                              "examples": [{"where": amvp.where[i].tolist(), "row": amvp.rows[i].tolist(), "got": got[i].tolist(), "want": amvp.mvp[i].tolist()} for i in bad[:3]]}
    else:
        report["amvp_cpu"] = {"derivations": 0, "mismatching": 0}
    # ---- cand_mode_list.hpp (what k_intra_gather makes a partition's most probable modes with) on the neighbour modes the encoder's CandModeList::getCandidate returned
    if len(intra):
        got = cpu.cand_mode_list(intra.mode_ab)
        want = np.concatenate([intra.ctx["cand_mode_list"].reshape(-1, 3), intra.ctx["neighbour_modes"].reshape(-1, 1)], axis=1).astype(np.int32)
        bad = np.flatnonzero(np.any(got != want, axis=1))
        top = intra.where[:, 2] % 64 == 0 if intra.where.shape[1] > 2 else np.zeros(len(intra), bool)
        report["cand_mode_list_cpu"] = {"partitions": int(len(intra)), "mismatching": int(len(bad)), "neighbours_differ": int((intra.mode_ab[:, 0] != intra.mode_ab[:, 1]).sum()),
                                        "angular_and_equal": int(((intra.mode_ab[:, 0] == intra.mode_ab[:, 1]) & (intra.mode_ab[:, 0] > 1)).sum()),
                                        "above_is_dc_at_the_top_of_a_ctu": bool((intra.mode_ab[top, 1] == 1).all()), "partitions_at_the_top_of_a_ctu": int(top.sum()),
                                        "examples": [{"ab": intra.mode_ab[i].tolist(), "got": got[i].tolist(), "want": want[i].tolist()} for i in bad[:3]]}
    else:
        report["cand_mode_list_cpu"] = {"partitions": 0, "mismatching": 0}
    # ---- picture_order.hpp: neighbourPositionAvailable (what the walk on the host and in k_search_rows decides its five reads by) against neighbourPuData's own three tests
    if len(amvp):
        got = cpu.positions_available(amvp.geometry)
        bad = np.flatnonzero(np.any(got != amvp.position, axis=1))
        report["availability_cpu"] = {"prediction_units": int(len(amvp)), "mismatching": int(len(bad)), "positions_not_available": int((amvp.position == 0).sum()),
                                      "by_position_A0_A1_B0_B1_B2": (amvp.position == 0).sum(axis=0).tolist(),
                                      "examples": [{"geometry": amvp.geometry[i].tolist(), "got": got[i].tolist(), "want": amvp.position[i].tolist()} for i in bad[:3]]}
    else:
        report["availability_cpu"] = {"prediction_units": 0, "mismatching": 0}
    # ---- merge.hpp (the reference's merge candidate list restated as data-only code) on the encoder's own neighbours: the list populateMergeCandidates left
    if len(merge):
        got = cpu.merge(merge.rows)
        bad = np.flatnonzero(np.any(got.reshape(len(got), -1) != merge.out.reshape(len(got), -1), axis=1))
        nb = merge.rows[:, 8:48].reshape(-1, 5, 8)
        avail = (nb[:, :, 0] != 0) | (nb[:, :, 1] != 0)
        bipred = (merge.out[:, :, 0] != 0) & (merge.out[:, :, 1] != 0)
        report["merge_cpu"] = {"derivations": int(len(merge)), "mismatching": int(len(bad)),
                               "with_the_temporal_candidate": int((merge.rows[:, 7] != 0).sum()),
                               "with_a_pruned_neighbour": int(np.sum(avail.sum(axis=1) > np.minimum(5, [len({tuple(x) for x, a in zip(n, av) if a}) for n, av in zip(nb, avail)]))),
                               "lists_with_a_bi_predictive_candidate": int(bipred.any(axis=1).sum()),
                               "second_units_of_a_split": int((merge.rows[:, 0] != 0).sum()),
                               "examples": [{"where": merge.where[i].tolist(), "row": merge.rows[i].tolist(), "got": got[i].tolist(), "want": merge.out[i].tolist()} for i in bad[:3]]}
    else:
        report["merge_cpu"] = {"derivations": 0, "mismatching": 0}
    # ---- amvp.hpp: deriveTemporalCandidate on the collocated picture's cells the encoder could read: the temporal candidates it derived (for its predictors and its merge lists)
    trows = np.concatenate([amvp.temporal, merge.temporal]) if len(amvp) or len(merge) else np.zeros((0, 36), np.int32)
    twant = np.concatenate([amvp.temporal_want, merge.temporal_want]) if len(trows) else np.zeros((0, 3), np.int32)
    if len(trows):
        got = cpu.temporal(trows)
        bad = np.flatnonzero(np.any(got != twant, axis=1))
        cells = trows[:, 14:34].reshape(-1, 2, 10)
        report["temporal_cpu"] = {"derivations": int(len(trows)), "for_predictors": int(len(amvp.temporal)), "for_merge_lists": int(len(merge.temporal)), "mismatching": int(len(bad)),
                                  "available": int((twant[:, 0] != 0).sum()), "from_a_cell_with_two_vectors": int(((cells[:, :, 0] != 0) & (cells[:, :, 1] != 0)).any(axis=1).sum()),
                                  "examples": [{"row": trows[i].tolist(), "got": got[i].tolist(), "want": twant[i].tolist()} for i in bad[:3]]}
    else:
        report["temporal_cpu"] = {"derivations": 0, "mismatching": 0}

    # ---- tu_decision.hpp on the encoder's own numbers (rates from its entropy estimator, distortions of three planes): the transform-tree decision
    # (Reconstruct.cpp:1296-1428) and the champion of an intra partition's RD refinement (Search.hpp:143-255)
    got = cpu.rqt_decide(rqt.rows) if len(rqt) else np.zeros((0, 2), np.int32)
    report["rqt_cpu"] = {"decisions_with_a_choice": int(len(rqt)), "units_left_uncoded_without_one": int(rqt.uncoded), "mismatching": int((got[:, 0] != rqt.chosen).sum()),
                         "chose_split": int((rqt.chosen == 1).sum())}
    champ = cpu.intra_rd_decide(intra.candidates, intra.count, intra.reciprocal_lambda) if len(intra) else np.zeros((0, 2), np.int32)
    report["intra_rd_cpu"] = {"partitions": int(len(intra)), "candidates": int(len(intra.candidates)), "rates_measured_by_the_encoder": int((intra.candidates[:, 2] >= 0).sum()),
                              "mismatching_champions": int((champ[:, 0] != intra.champion).sum()),
                              "champion_is_not_the_first_candidate": int((champ[:, 1] != 0).sum())}

    # ---- the product: batch clients (and, on the MI355X, the kernels that hold the loops) on the same inputs
    if args.device != "none":
        dev = Device(args.device)
        origin = lambda stride: tt.PAD * stride + tt.PAD
        lim = args.limit or 1 << 30
        r = {"searches": 0, "mismatching_launch_and_replay": 0, "launches": 0, "rounds": 0}
        if args.device == "real":
            r["mismatching_loops_in_kernel"] = 0
        t0 = time.perf_counter()
        budget = lim
        for key, idx in uni.groups().items():
            idx = idx[:budget]
            budget -= len(idx)
            if not len(idx):
                break
            par = uni.params(key, w, h)
            (s, stride), (rf, _) = P(padded_src, src, key[0]), P(padded_rec, recon, key[1])
            S = s.itemsize
            d_src, d_ref = dev.plane("src", key[0], s), dev.plane("ref", key[1], rf)
            d_phase, pe = dev.phase(key[1], rf, stride, w, h, key[4])
            for lst in (0, 1):      # the batch client takes one list's searches per call (one reference picture)
                sel = idx[uni.pus["ref_list"][idx] == lst]
                if not len(sel):
                    continue
                pus = np.ascontiguousarray(uni.pus[sel])
                exp = uni.results[sel]
                out = np.zeros(len(sel), st.RESULT_DT)
                stats = Stats()
                rc = dev.L.havoc_search_motion_uni(dev.ctx, S, C.byref(par), d_src, origin(stride), stride, d_ref, origin(stride), stride, tt.PAD, d_phase, pe,
                                                   origin(stride), pus.ctypes.data, len(pus), out.ctypes.data, 8, C.byref(stats))
                assert rc == 0, (rc, dev.dev.havoc_mi355x_last_error())
                r["mismatching_launch_and_replay"] += int(len(differing(out, exp, UNI_FIELDS)))
                r["launches"] += stats.launches
                r["rounds"] = max(r["rounds"], stats.rounds)
                if args.device == "real":
                    out = np.zeros(len(sel), st.RESULT_DT)
                    rc = dev.L.havoc_search_motion_uni_device(dev.ctx, S, C.byref(par), d_src, origin(stride), stride, d_ref, origin(stride), stride, tt.PAD, d_phase,
                                                              pe, origin(stride), pus.ctypes.data, len(pus), out.ctypes.data, C.byref(stats))
                    assert rc == 0, (rc, dev.dev.havoc_mi355x_last_error())
                    bad = differing(out, exp, UNI_FIELDS)
                    r["mismatching_loops_in_kernel"] += int(len(bad))
                    if len(bad) and "examples" not in r:
                        r["examples"] = [{"pu": [int(v) for v in (pus[k]["x0"], pus[k]["y0"], pus[k]["w"], pus[k]["h"])], "mvp": pus[k]["mvp"].tolist(),
                                          "got": {f: np.asarray(out[k][f]).tolist() for f in UNI_FIELDS + ["cost_subpel"]}, "want": {f: np.asarray(exp[k][f]).tolist() for f in UNI_FIELDS},
                                          "cpu_cost_subpel": int(expected_uni[key][1][np.flatnonzero(expected_uni[key][0] == sel[k])[0]]["cost_subpel"])} for k in bad[:6]]
                r["searches"] += len(sel)
        r["seconds"] = round(time.perf_counter() - t0, 2)
        report["uni_device"] = r

        r = {"searches": 0, "mismatching_launch_and_replay": 0}
        if args.device == "real":
            r["mismatching_loops_in_kernel"] = 0
        t0 = time.perf_counter()
        budget = lim
        for key, par, lst, other, sel, pus, start in bi_jobs:
            sel, pus, start = sel[:budget], np.ascontiguousarray(pus[:budget]), np.ascontiguousarray(start[:budget])
            budget -= len(sel)
            if not len(sel):
                break
            (s, stride), (rf, _), (ro, _) = P(padded_src, src, key[0]), P(padded_rec, recon, key[1]), P(padded_rec, recon, other)
            S = s.itemsize
            d_src, d_ref, d_other = dev.plane("src", key[0], s), dev.plane("ref", key[1], rf), dev.plane("ref", other, ro)
            d_phase, pe = dev.phase(key[1], rf, stride, w, h, key[4])
            exp = bi.results[sel]
            out = np.zeros(len(sel), st.RESULT_DT)
            stats = Stats()
            rc = dev.L.havoc_search_motion_bi(dev.ctx, S, C.byref(par), d_src, origin(stride), stride, d_ref, origin(stride), stride, tt.PAD, d_phase, pe, origin(stride),
                                              d_other, origin(stride), pus.ctypes.data, start.ctypes.data, len(pus), out.ctypes.data, 8, C.byref(stats))
            assert rc == 0, (rc, dev.dev.havoc_mi355x_last_error())
            r["mismatching_launch_and_replay"] += int(len(differing(out, exp, BI_FIELDS)))
            if args.device == "real":
                d_phase_other, _ = dev.phase(other, ro, stride, w, h, key[4])
                out = np.zeros(len(sel), st.RESULT_DT)
                rc = dev.L.havoc_search_motion_bi_device(dev.ctx, S, C.byref(par), d_src, origin(stride), stride, d_ref, origin(stride), stride, tt.PAD, d_phase, pe,
                                                         origin(stride), d_phase_other, origin(stride), pus.ctypes.data, start.ctypes.data, len(pus), out.ctypes.data,
                                                         C.byref(stats))
                assert rc == 0, (rc, dev.dev.havoc_mi355x_last_error())
                r["mismatching_loops_in_kernel"] += int(len(differing(out, exp, BI_FIELDS)))
            r["searches"] += len(sel)
        r["seconds"] = round(time.perf_counter() - t0, 2)
        report["bi_device"] = r

        # the mode order taken on the device (k_intra_order; the stand-in has its own plain-C version of the kernel)
        r = {"partitions": 0, "mismatching": 0}
        MAXO = 12
        for rsl in np.unique(intra.rsl):
            sel = np.flatnonzero((intra.rsl == rsl) & (intra.ctx["max_refine"] + intra.ctx["neighbour_modes"] <= MAXO))[:lim]
            n = len(sel)
            if not n:
                continue
            lam = np.zeros(1, np.int32)
            lam[0] = int(rsl * 65536 + 0.5)
            satd, mpm = np.ascontiguousarray(intra.satd[sel]), np.ascontiguousarray(intra.ctx[sel])
            d_satd, d_mpm = dev._alloc(satd.nbytes), dev._alloc(mpm.nbytes)
            d_order, d_count, d_slot, d_total = dev._alloc(4 * MAXO * n), dev._alloc(4 * n), dev._alloc(4 * n), dev._alloc(16)
            assert dev.dev.havoc_mi355x_h2d(dev.ctx, d_satd, satd.ctypes.data, satd.nbytes) == 0 and dev.dev.havoc_mi355x_h2d(dev.ctx, d_mpm, mpm.ctypes.data, mpm.nbytes) == 0
            assert dev.dev.havoc_mi355x_intra_order(dev.ctx, d_satd, d_mpm, n, int(lam[0]), d_order, d_count, d_slot, d_total) == 0, dev.dev.havoc_mi355x_last_error()
            order, count = np.zeros((n, MAXO), np.int32), np.zeros(n, np.int32)
            assert dev.dev.havoc_mi355x_d2h(dev.ctx, order.ctypes.data, d_order, order.nbytes) == 0 and dev.dev.havoc_mi355x_d2h(dev.ctx, count.ctypes.data, d_count, count.nbytes) == 0
            bad = (count != intra.count[sel]) | np.any((order != intra.order[sel][:, :MAXO]) & (np.arange(MAXO)[None, :] < intra.count[sel][:, None]), axis=1)
            r["mismatching"] += int(bad.sum())
            r["partitions"] += n
        report["intra_device"] = r
        report["intra_neighbours_device"] = intra_neighbours(intra, w, h, internal, dev)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
