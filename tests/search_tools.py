"""Test-side tooling for the decision-loop clients (turingcodec_amd/search): builds tests/search_client.cpp against the
three back ends, generates a seeded list of (prediction unit, list) searches with predictors on the synthetic clip of
SURVEY.md 8(d), and runs the clients through ctypes.

    ref      tests/_build/libsearch_ref.so      the reference's own havoc library (oracle/_ref) behind the table API
    oracle   tests/_build/libsearch_oracle.so   the CPU oracle (no table API)
    classic  tests/_build/libsearch_classic.so  libhavoc_classic.so = the MI355X implementation behind the table API
"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tests", "_build")
SRC = os.path.join(ROOT, "tests", "search_client.cpp")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "turingcodec_amd", "search"), "-I" + os.path.join(ROOT, "tests")]

PU_DT = np.dtype([("x0", "i4"), ("y0", "i4"), ("w", "i4"), ("h", "i4"), ("cu_log2_size", "i4"), ("cqt_depth", "i4"), ("part_2Nx2N", "i4"),
                  ("ref_list", "i4"), ("x_ctb", "i4"), ("y_ctb", "i4"), ("mvp", "i2", (2, 2)), ("mv_previous_2Nx2N", "i2", (2,)),
                  ("mv_other", "i2", (2,)), ("mvp_rate", "i8", (2,))])
RESULT_DT = np.dtype([("mv", "i2", (2,)), ("mvd", "i2", (2,)), ("mv_integer", "i2", (2,)), ("mvp_flag", "i2"), ("wrote_2Nx2N", "i2"),
                      ("calls", "i4"), ("replays", "i4"), ("cost_integer", "i8"), ("cost_subpel", "i8"), ("cost_mvd_zero", "i8", (2,))])
INTRA_CTX_DT = np.dtype([("cand_mode_list", "i4", (3,)), ("neighbour_modes", "i4"), ("max_refine", "i4"), ("reserved", "i4"),
                         ("rate_a_minus_c", "i8"), ("rate_b_minus_c", "i8")])
INTRA_RESULT_DT = np.dtype([("costs", "i8", (35,)), ("order", "i4", (35,)), ("count", "i4")])
assert PU_DT.itemsize == 72 and RESULT_DT.itemsize == 56 and INTRA_CTX_DT.itemsize == 40 and INTRA_RESULT_DT.itemsize == 424


from turingcodec_amd.decisions import SearchParams as Params  # noqa: E402  (havoc_search_params; one definition for product binding and tests)


def reciprocal_sqrt_lambda(qp, qp_factor=0.68, non_reference=True):
    """computeLambda, turing/Measure.h:58-78 (B picture of the hierarchy), then 1 / sqrt"""
    lam = qp_factor * 2.0 ** ((qp - 12.0) / 3.0)
    if non_reference:
        lam *= min(4.0, max(2.0, (qp - 12.0) / 6.0))
    return 1.0 / math.sqrt(lam)


def medium_params(width, height, bit_depth=8, qp=32, concurrent_frames=4):
    return Params(width, height, 64, concurrent_frames, 1, 0, 0, 1, 1, bit_depth, reciprocal_sqrt_lambda(qp))


def _newer(target, *deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(kind):
    """compile the client for one back end; returns the library path (None when the back end's library is not there)"""
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, f"libsearch_{kind}.so")
    hdrs = [os.path.join(ROOT, "turingcodec_amd", "search", f) for f in ("decision.hpp", "table_view.hpp", "search_abi.h", "picture_order.hpp", "amvp.hpp", "cand_mode_list.hpp", "tu_decision.hpp")]
    hdrs.append(os.path.join(ROOT, "tests", "merge.hpp"))      # test infrastructure since round 6: pinned, but nothing in the product derives a merge list
    base = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-Wall"] + INC + [SRC, "-o", out]
    if kind == "ref":
        lib = os.path.join(ROOT, "oracle", "_ref", "libhavoc_ref.so")
        if not os.path.exists(lib):
            return out if os.path.exists(out) else None
        if not _newer(out, SRC, lib, *hdrs):
            subprocess.check_call(base + ["-L" + os.path.dirname(lib), "-lhavoc_ref", "-Wl,-rpath,$ORIGIN/../../oracle/_ref"])
    elif kind == "oracle":
        lib = os.path.join(ROOT, "oracle", "liboracle.so")
        if not _newer(out, SRC, lib, *hdrs):
            subprocess.check_call(base + ["-DSEARCH_ORACLE", "-L" + os.path.dirname(lib), "-loracle", "-Wl,-rpath,$ORIGIN/../../oracle"])
    elif kind == "classic":
        lib = os.path.join(ROOT, "turingcodec_amd", "libhavoc_classic.so")
        if not os.path.exists(lib):
            return None
        if not _newer(out, SRC, lib, *hdrs, os.path.join(ROOT, "include", "havoc_classic_ext.h")):
            subprocess.check_call(base + ["-DHAVOC_CLASSIC_EXT", "-L" + os.path.dirname(lib), "-lhavoc_classic", "-lhavoc_mi355x",
                                          "-Wl,-rpath,$ORIGIN/../../turingcodec_amd"])
    else:
        raise ValueError(kind)
    return out


class Client:
    def __init__(self, kind, mask=3):
        path = build(kind)
        if path is None:
            raise FileNotFoundError(kind)
        self.kind = kind
        L = self.L = C.CDLL(path)
        vp, ip, i = C.c_void_p, C.c_ssize_t, C.c_int
        L.client_open.argtypes = [i]
        L.client_uni.argtypes = [i, vp, ip, vp, ip, C.POINTER(Params), vp, i, i, vp]
        L.client_bi.argtypes = [i, vp, ip, vp, vp, ip, C.POINTER(Params), vp, vp, i, i, vp]
        L.client_uni_lanes.argtypes = [i, vp, ip, vp, ip, C.POINTER(Params), vp, i, i, vp]
        L.client_uni_logged.argtypes = [i, vp, ip, vp, ip, C.POINTER(Params), vp, i, i, vp, vp, C.c_int64, vp]
        L.client_uni_logged.restype = C.c_int64
        L.client_bi_logged.argtypes = [i, vp, ip, vp, vp, ip, C.POINTER(Params), vp, vp, i, i, vp, vp, C.c_int64, vp]
        L.client_bi_lanes.argtypes = [i, vp, ip, vp, vp, ip, C.POINTER(Params), vp, vp, i, i, vp]
        L.client_bi_logged.restype = C.c_int64
        L.client_rqt_decide.argtypes = [vp, i, vp]
        L.client_amvp.argtypes = [vp, i, vp]
        L.client_merge.argtypes = [vp, i, vp]
        L.client_temporal.argtypes = [vp, i, vp]
        L.client_cand_mode_list.argtypes = [vp, i, vp]
        L.client_positions_available.argtypes = [vp, i, vp]
        L.client_check_lds_neighbours.argtypes = [vp, vp, i, i, i, i, i, vp]
        L.client_intra_rd_decide.argtypes = [vp, vp, vp, i, vp]
        L.client_intra_order.argtypes = [vp, C.c_double, vp, i, vp]
        L.client_intra35.argtypes = [i, i, i, vp, ip, vp, vp, i, vp]
        L.client_picture_uni.argtypes = [i, vp, ip, vp, vp, ip, C.POINTER(Params), vp, vp, i, i, vp, vp, vp, vp]
        if kind != "classic":
            L.client_rqt.argtypes = [i, i, vp, ip, vp, ip, vp, ip, vp, vp, C.c_double, C.c_double, i, vp, i, vp]
            L.client_intra_rd.argtypes = [i, i, i, vp, ip, vp, vp, i, vp, vp, vp, vp, vp, C.c_double, C.c_double, i, vp, vp]
        if kind == "classic":
            L.client_register.argtypes = [vp, ip, i, i, i, i, i, i]
            L.client_unregister.argtypes = [vp]
            L.client_stats.argtypes = [vp]
        assert L.client_open(mask) == 0

    @staticmethod
    def _origin(plane, stride, pad):
        return plane.ctypes.data + (pad * stride + pad) * plane.itemsize

    def uni(self, params, src, ref, stride, pad, pus, b=0, e=None):
        """src / ref: padded planes (flat numpy arrays, `stride` samples per row, `pad` samples of border)"""
        e = len(pus) if e is None else e
        out = np.zeros(len(pus), RESULT_DT)
        rc = self.L.client_uni(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), stride, C.byref(params),
                               pus.ctypes.data, b, e, out.ctypes.data)
        assert rc == 0
        return out

    def uni_lanes(self, params, src, ref, stride, pad, pus):
        """uni() through decision.hpp's step hooks in the device view's formulation (a candidate per lane, smallest key wins), emulated on the host"""
        out = np.zeros(len(pus), RESULT_DT)
        assert self.L.client_uni_lanes(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), stride, C.byref(params),
                                       pus.ctypes.data, 0, len(pus), out.ctypes.data) == 0
        return out

    def uni_logged(self, params, src, ref, stride, pad, pus, capacity):
        """uni() + the log of every View call the loops made: (results, rows int32 [calls, 13], first int64 [len(pus) + 1])"""
        out = np.zeros(len(pus), RESULT_DT)
        rows = np.zeros((capacity, 13), np.int32)
        first = np.zeros(len(pus) + 1, np.int64)
        n = self.L.client_uni_logged(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), stride, C.byref(params),
                                     pus.ctypes.data, 0, len(pus), out.ctypes.data, rows.ctypes.data, capacity, first.ctypes.data)
        assert 0 <= n <= capacity, (n, capacity)
        return out, rows[:n], first

    def bi_logged(self, params, src, ref, ref_other, stride, pad, pus, start, capacity):
        out = np.zeros(len(pus), RESULT_DT)
        rows = np.zeros((capacity, 13), np.int32)
        first = np.zeros(len(pus) + 1, np.int64)
        start = np.ascontiguousarray(start, np.int16)
        n = self.L.client_bi_logged(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), self._origin(ref_other, stride, pad),
                                    stride, C.byref(params), pus.ctypes.data, start.ctypes.data, 0, len(pus), out.ctypes.data, rows.ctypes.data, capacity,
                                    first.ctypes.data)
        assert 0 <= n <= capacity, (n, capacity)
        return out, rows[:n], first

    def bi_lanes(self, params, src, ref, ref_other, stride, pad, pus, start):
        """bi() with the exhaustive grid in the device view's formulation (a candidate per lane, smallest (cost, index) key), emulated on the host"""
        out = np.zeros(len(pus), RESULT_DT)
        start = np.ascontiguousarray(start, np.int16)
        assert self.L.client_bi_lanes(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), self._origin(ref_other, stride, pad),
                                      stride, C.byref(params), pus.ctypes.data, start.ctypes.data, 0, len(pus), out.ctypes.data) == 0
        return out

    def rqt_decide(self, rows):
        """tu_decision.hpp: decideRqt on recorded (cbf, weighted ssd, rate) of the split tree and (ssd, rate) of the unsplit block: int32 [n, 2] = depth, tried_zero"""
        rows = np.ascontiguousarray(rows, np.int64)
        out = np.zeros((len(rows), 2), np.int32)
        assert self.L.client_rqt_decide(rows.ctypes.data, len(rows), out.ctypes.data) == 0
        return out

    def cand_mode_list(self, ab):
        """turingcodec_amd/search/cand_mode_list.hpp: candModeListOf on recorded (A, B) (int32 [n, 2]) -> int32 [n, 4] = candModeList[0..2], neighbourModes"""
        ab = np.ascontiguousarray(ab, np.int32)
        out = np.zeros((len(ab), 4), np.int32)
        assert self.L.client_cand_mode_list(ab.ctypes.data, len(ab), out.ctypes.data) == 0
        return out

    def positions_available(self, rows):
        """turingcodec_amd/search/picture_order.hpp: neighbourPositionAvailable for A0, A1, B0, B1, B2 of prediction units (int32 [n, 8]: x0, y0, w, h, ctb, picture width, height, 0)"""
        rows = np.ascontiguousarray(rows, np.int32)
        out = np.zeros((len(rows), 5), np.int32)
        assert self.L.client_positions_available(rows.ctypes.data, len(rows), out.ctypes.data) == 0
        return out

    def temporal(self, rows):
        """turingcodec_amd/search/amvp.hpp: deriveTemporalCandidate on recorded inputs (int32 [n, 36]) -> int32 [n, 3] = available, x, y"""
        rows = np.ascontiguousarray(rows, np.int32)
        out = np.zeros((len(rows), 3), np.int32)
        assert self.L.client_temporal(rows.ctypes.data, len(rows), out.ctypes.data) == 0
        return out

    def merge(self, rows):
        """tests/merge.hpp: deriveMergeCandidates on recorded inputs (int32 [n, 64]) -> int32 [n, 5, 8] = per candidate predFlag0, predFlag1, refIdx0,
        refIdx1, mv0.x, mv0.y, mv1.x, mv1.y"""
        rows = np.ascontiguousarray(rows, np.int32)
        out = np.zeros((len(rows), 5, 8), np.int32)
        assert self.L.client_merge(rows.ctypes.data, len(rows), out.ctypes.data) == 0
        return out

    def amvp(self, rows):
        """turingcodec_amd/search/amvp.hpp: deriveAmvp on recorded inputs (int32 [n, 52]) -> int32 [n, 4] = mvp[0].x, .y, mvp[1].x, .y"""
        rows = np.ascontiguousarray(rows, np.int32)
        out = np.zeros((len(rows), 4), np.int32)
        assert self.L.client_amvp(rows.ctypes.data, len(rows), out.ctypes.data) == 0
        return out

    def check_lds_neighbours(self, pus, ctu_first, ctus_x, ctus_y, pic_w, pic_h, ctb=64):
        """the device walk's view of a CTU's neighbours (csrc/kernels_search.hip: 256 cells of the CTU + 16 left + 16 above + the two corners, load_neighbours' addressing,
        restated on the host) against the whole motion field: how many derivations differ, and the first one (pu, list, field's two predictors, view's two)"""
        pus, ctu_first = np.ascontiguousarray(pus), np.ascontiguousarray(ctu_first, np.int32)
        example = np.zeros(6, np.int32)
        return self.L.client_check_lds_neighbours(pus.ctypes.data, ctu_first.ctypes.data, ctus_x, ctus_y, pic_w, pic_h, ctb, example.ctypes.data), example

    def intra_rd_decide(self, cand, count, rl):
        """tu_decision.hpp: decideIntraRd on recorded (mode, ssd, rate or -1) per candidate: int32 [n, 2] = champion's mode, its index"""
        cand, count, rl = np.ascontiguousarray(cand, np.int64), np.ascontiguousarray(count, np.int32), np.ascontiguousarray(rl, np.int32)
        out = np.zeros((len(count), 2), np.int32)
        assert self.L.client_intra_rd_decide(cand.ctypes.data, count.ctypes.data, rl.ctypes.data, len(count), out.ctypes.data) == 0
        return out

    def picture_uni(self, params, src, ref0, ref1, stride, pad, pus, ctu_first, ctus_x, ctus_y, mvp_rate=(65536, 65536), bi=False):
        """a whole picture's searches in dependency order, one table call at a time (turingcodec_amd/search/picture_order.hpp):
        (results [2 * len(pus)], field int16 [2, cells_y, cells_x, 2]); with bi also the bi-directional refinements [2 * len(pus)] as a third value"""
        out = np.zeros(2 * len(pus), RESULT_DT)
        out_bi = np.zeros(2 * len(pus), RESULT_DT) if bi else None
        field = np.zeros((2, (params.pic_height + 3) // 4, (params.pic_width + 3) // 4, 2), np.int16)
        rate = np.asarray(mvp_rate, np.int64)
        pus = np.ascontiguousarray(pus)
        ctu_first = np.ascontiguousarray(ctu_first, np.int32)
        rc = self.L.client_picture_uni(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref0, stride, pad), self._origin(ref1, stride, pad), stride,
                                       C.byref(params), pus.ctypes.data, ctu_first.ctypes.data, ctus_x, ctus_y, rate.ctypes.data, out.ctypes.data, field.ctypes.data,
                                       out_bi.ctypes.data if bi else None)
        assert rc == 0
        return (out, field, out_bi) if bi else (out, field)

    def rqt(self, bit_depth, src, stride, pad, pred, pred_stride, states, quant, lam, reciprocal_lambda, cus, sdh=1):
        """the residual-quadtree decisions one block at a time (turingcodec_amd/search/tu_decision.hpp) through this back end's primitives and
        Rdoq: (results RQT_RESULT_DT, reconstruction plane like src)"""
        from turingcodec_amd.decisions import RQT_RESULT_DT
        out = np.zeros(len(cus), RQT_RESULT_DT)
        rec = np.zeros_like(src)
        quant = np.ascontiguousarray(quant, np.int32)
        states = np.ascontiguousarray(states, np.uint8)
        cus = np.ascontiguousarray(cus)
        rc = self.L.client_rqt(src.itemsize, bit_depth, self._origin(src, stride, pad), stride, pred.ctypes.data, pred_stride, self._origin(rec, stride, pad), stride,
                               states.ctypes.data, quant.ctypes.data, float(lam), float(reciprocal_lambda), int(sdh), cus.ctypes.data, len(cus), out.ctypes.data)
        assert rc == 0
        return out, rec

    def intra_rd(self, bit_depth, log2, src, stride, nb, jobs, order, ictx, ctx_index, states, quant_row, lam, reciprocal_lambda, sdh=1):
        """the RD refinement of intra partitions one candidate at a time (tu_decision.hpp: decideIntraRd): (INTRA_RD_RESULT_DT[n], reconstructions [n, N*N])"""
        from turingcodec_amd.decisions import INTRA_RD_RESULT_DT
        jobs = np.ascontiguousarray(jobs, np.int32)
        n, area = len(jobs), 1 << 2 * log2
        out = np.zeros(n, INTRA_RD_RESULT_DT)
        rec = np.zeros((n, area), src.dtype)
        order, ictx = np.ascontiguousarray(order), np.ascontiguousarray(ictx)
        ctx_index = np.ascontiguousarray(ctx_index, np.int32)
        states = np.ascontiguousarray(states, np.uint8)
        quant_row = np.ascontiguousarray(quant_row, np.int32)
        rc = self.L.client_intra_rd(src.itemsize, bit_depth, log2, src.ctypes.data, stride, nb.ctypes.data, jobs.ctypes.data, n, order.ctypes.data, ictx.ctypes.data,
                                    ctx_index.ctypes.data, states.ctypes.data, quant_row.ctypes.data, float(lam), float(reciprocal_lambda), int(sdh),
                                    rec.ctypes.data, out.ctypes.data)
        assert rc == 0
        return out, rec

    def bi(self, params, src, ref, ref_other, stride, pad, pus, start):
        out = np.zeros(len(pus), RESULT_DT)
        start = np.ascontiguousarray(start, np.int16)
        rc = self.L.client_bi(src.itemsize, self._origin(src, stride, pad), stride, self._origin(ref, stride, pad), self._origin(ref_other, stride, pad),
                              stride, C.byref(params), pus.ctypes.data, start.ctypes.data, 0, len(pus), out.ctypes.data)
        assert rc == 0
        return out

    def intra35(self, bit_depth, log2, src, stride, nb, jobs):
        """per-call 35-mode SATD stage; src = flat plane array addressed by the jobs' src_off, jobs int32 [n, 8]"""
        jobs = np.ascontiguousarray(jobs, np.int32)
        out = np.zeros((len(jobs), 35), np.int32)
        assert self.L.client_intra35(src.itemsize, bit_depth, log2, src.ctypes.data, stride, nb.ctypes.data, jobs.ctypes.data, len(jobs), out.ctypes.data) == 0
        return out

    def intra_order(self, ctx, rsl, satd35):
        satd35 = np.ascontiguousarray(satd35, np.int32)
        out = np.zeros(len(ctx), INTRA_RESULT_DT)
        assert self.L.client_intra_order(ctx.ctypes.data, rsl, satd35.ctypes.data, len(ctx), out.ctypes.data) == 0
        return out

    # classic only
    def register(self, plane, stride, pad, width, height, bit_depth, role):
        return self.L.client_register(self._origin(plane, stride, pad), stride, width, height, pad, plane.itemsize, bit_depth, role)

    def unregister(self, plane, stride, pad):
        return self.L.client_unregister(self._origin(plane, stride, pad))

    def stats(self):
        a = (C.c_int64 * 8)()
        self.L.client_stats(a)
        return [int(v) for v in a]


# (PartMode name, [(dx, dy, w, h) in units of the CU size / 4]) -- the part modes of turing/Search.hpp's inter loop
PART_MODES = [("2Nx2N", [(0, 0, 4, 4)]), ("2NxN", [(0, 0, 4, 2), (0, 2, 4, 2)]), ("Nx2N", [(0, 0, 2, 4), (2, 0, 2, 4)]),
              ("2NxnU", [(0, 0, 4, 1), (0, 1, 4, 3)]), ("2NxnD", [(0, 0, 4, 3), (0, 3, 4, 1)]), ("nLx2N", [(0, 0, 1, 4), (1, 0, 3, 4)]),
              ("nRx2N", [(0, 0, 3, 4), (3, 0, 1, 4)])]


def make_searches(width, height, n, seed, hard_fraction=0.15):
    """n (PU, list) searches on the clip of workload.synth_frames: coding units of 8..64 at aligned positions, every part
    mode the inter loop tries, predictors = the clip's true motion ((3,2) samples per frame towards list 0) plus noise;
    `hard_fraction` of them get predictors far off (long star / raster searches) or sit on the picture border (limits)."""
    rng = np.random.default_rng(seed)
    pus = np.zeros(n, PU_DT)
    for i in range(n):
        log2 = int(rng.choice([3, 4, 5, 6], p=[0.35, 0.35, 0.2, 0.1]))
        size = 1 << log2
        modes = PART_MODES[:3] if log2 == 3 else PART_MODES      # no AMP on 8x8 coding units
        name, parts = modes[int(rng.integers(0, len(modes)))]
        dx, dy, w4, h4 = parts[int(rng.integers(0, len(parts)))]
        q = size // 4
        edge = rng.random() < hard_fraction / 2
        cx = (int(rng.integers(0, (width - size) // size + 1)) * size) if not edge else int(rng.choice([0, (width - size) // size * size]))
        cy = (int(rng.integers(0, (height - size) // size + 1)) * size) if not edge else int(rng.choice([0, (height - size) // size * size]))
        p = pus[i]
        p["x0"], p["y0"], p["w"], p["h"] = cx + dx * q, cy + dy * q, w4 * q, h4 * q
        p["cu_log2_size"], p["cqt_depth"], p["part_2Nx2N"] = log2, 6 - log2, int(name == "2Nx2N")
        p["ref_list"] = int(rng.integers(0, 2))
        p["x_ctb"], p["y_ctb"] = cx & ~63, cy & ~63
        true = np.array([12, 8]) * (1 if p["ref_list"] == 0 else -1)
        far = rng.random() < hard_fraction
        spread = 160 if far else 6
        p["mvp"][0] = true + rng.integers(-spread, spread + 1, 2)
        p["mvp"][1] = true + rng.integers(-3 * spread, 3 * spread + 1, 2)
        p["mv_previous_2Nx2N"] = ((true + rng.integers(-8, 9, 2)) // 4) * 4
        p["mv_other"] = -true + rng.integers(-5, 6, 2)
        p["mvp_rate"] = rng.integers(30000, 140000, 2)
    return pus


def clip_planes(width, height, seed, bit_depth=8, pad=96, distance=1):
    """(src, ref L0, ref L1, stride): padded luma planes of three frames of the synthetic clip, the references `distance` frames before and after
    the source (1: consecutive frames; 4, 8: the long vectors of the hierarchy's upper layers -- star search and raster refinement)"""
    from turingcodec_amd.workload import pad_plane, synth_frames
    frames = synth_frames(width, height, 2 * distance + 1, seed, bit_depth)
    planes = [pad_plane(f[0], pad) for f in (frames[distance], frames[0], frames[2 * distance])]
    stride = planes[0].shape[1]
    return [np.ascontiguousarray(p.ravel()) for p in planes], stride


def make_intra_contexts(n, log2, seed):
    """per-partition state of the 35-mode stage: most probable modes, rate offsets (Search.hpp:55-87), refinement count at medium"""
    rng = np.random.default_rng(seed)
    ctx = np.zeros(n, INTRA_CTX_DT)
    for i in range(n):
        ctx[i]["cand_mode_list"] = rng.choice(35, 3, replace=False)
        ctx[i]["neighbour_modes"] = 3
        ctx[i]["max_refine"] = 3 if log2 > 3 else 8      # Speed::nCandidatesIntraRefinement at medium
        ctx[i]["rate_a_minus_c"] = -int(rng.integers(300000, 420000))
        ctx[i]["rate_b_minus_c"] = -int(rng.integers(100000, 200000))
    return ctx
