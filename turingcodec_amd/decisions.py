"""Host-side binding of libhavoc_search.so's PICTURE client (turingcodec_amd/search/picture_search.cpp): a whole picture's
uni-directional motion searches in wavefront order with predictors derived from earlier decisions (search/picture_order.hpp).

Plain ctypes over raw device pointers and a havoc_mi355x context handle, so that the same binding drives the real library (bench.py,
-m gpu tests) and the CPU stand-in device of the host-logic tests.  There is no CPU path here: the library it loads launches kernels.
"""
import ctypes as C
import os

import numpy as np

from .workload import PICTURE_PU_DT

_HERE = os.path.dirname(os.path.abspath(__file__))
SEARCH_LIB = os.path.join(_HERE, "libhavoc_search.so")

RESULT_DT = np.dtype([("mv", "i2", (2,)), ("mvd", "i2", (2,)), ("mv_integer", "i2", (2,)), ("mvp_flag", "i2"), ("wrote_2Nx2N", "i2"),
                      ("calls", "i4"), ("replays", "i4"), ("cost_integer", "i8"), ("cost_subpel", "i8"), ("cost_mvd_zero", "i8", (2,))])
assert RESULT_DT.itemsize == 56 and PICTURE_PU_DT.itemsize == 32


class SearchParams(C.Structure):
    """havoc_search_params (search/search_abi.h): what the loops read from encoder state"""
    _fields_ = [("pic_width", C.c_int32), ("pic_height", C.c_int32), ("ctb_size", C.c_int32), ("concurrent_frames", C.c_int32),
                ("met", C.c_int32), ("small_search_window", C.c_int32), ("bi_small_search_window", C.c_int32), ("half_pel", C.c_int32),
                ("quarter_pel", C.c_int32), ("bit_depth", C.c_int32), ("reciprocal_sqrt_lambda", C.c_double)]


class PictureStats(C.Structure):
    """havoc_picture_stats"""
    _fields_ = [("steps", C.c_int32), ("rounds", C.c_int32), ("max_rounds_in_step", C.c_int32), ("launches", C.c_int32), ("surfaces_small", C.c_int32),
                ("surfaces_zero", C.c_int32), ("surfaces_large", C.c_int32), ("satd_jobs", C.c_int32), ("speculative_runs", C.c_int32), ("reruns", C.c_int32),
                ("bytes_down", C.c_int64), ("seconds_gpu", C.c_double), ("seconds_host", C.c_double), ("seconds_total", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


RQT_CU_DT = np.dtype([("x0", "i4"), ("y0", "i4"), ("log2_size", "i4"), ("ctx_index", "i4")])                       # havoc_rqt_cu
TU_OUTCOME_DT = np.dtype([("cbf", "i4"), ("ssd", "u4"), ("nonzero", "i4"), ("sum_abs", "i4")])                    # havoc_tu_outcome
RQT_RESULT_DT = np.dtype([("depth", "i4"), ("tried_zero", "i4"), ("zero", TU_OUTCOME_DT), ("one", TU_OUTCOME_DT, (4,)), ("cost_zero", "i8"),
                          ("cost_one", "i8")])                                                                     # havoc_rqt_result
assert RQT_CU_DT.itemsize == 16 and RQT_RESULT_DT.itemsize == 104


INTRA_CTX_DT = np.dtype([("cand_mode_list", "i4", (3,)), ("neighbour_modes", "i4"), ("max_refine", "i4"), ("reserved", "i4"), ("rate_a_minus_c", "i8"),
                         ("rate_b_minus_c", "i8")])                                                               # havoc_search_intra_ctx
INTRA_RESULT_DT = np.dtype([("costs", "i8", (35,)), ("order", "i4", (35,)), ("count", "i4")])                       # havoc_search_intra_result
INTRA_RD_RESULT_DT = np.dtype([("mode", "i4"), ("index", "i4"), ("evaluated", "i4"), ("reserved", "i4"), ("cost", "i8"), ("outcome", TU_OUTCOME_DT)])
assert INTRA_CTX_DT.itemsize == 40 and INTRA_RESULT_DT.itemsize == 424 and INTRA_RD_RESULT_DT.itemsize == 40


class RqtStats(C.Structure):
    _fields_ = [("launches", C.c_int32), ("candidates", C.c_int32), ("seconds_gpu", C.c_double), ("seconds_host", C.c_double), ("seconds_total", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def rqt_quant(qp, bit_depth):
    """havoc_rqt_quant[4] for transform sizes 4..32 of an inter (non-I slice) luma block (turing/QpState.h:85-94, Reconstruct.cpp:774-782)"""
    from . import workload
    q = np.zeros((4, 4), np.int32)
    for log2 in (2, 3, 4, 5):
        qs, qshift, _ = workload.quant_params(qp, log2, bit_depth, False)
        inv, dshift = workload.dequant_params(qp, log2, bit_depth)
        q[log2 - 2] = (qs, qshift, inv, dshift)
    return q


def medium_params(width, height, bit_depth, reciprocal_sqrt_lambda, concurrent_frames=4):
    """speed=medium: multiple early termination on, full search window, half- and quarter-sample refinement (turing/Speed.h)"""
    return SearchParams(width, height, 64, concurrent_frames, 1, 0, 0, 1, 1, bit_depth, reciprocal_sqrt_lambda)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SEARCH_LIB):
            raise RuntimeError(f"{SEARCH_LIB} is missing: build it with `make -C turingcodec_amd/csrc` -- there is no fallback path")
        L = C.CDLL(SEARCH_LIB)
        vp, ip, i64 = C.c_void_p, C.c_ssize_t, C.c_int64
        L.havoc_search_picture_uni.argtypes = [vp, C.c_int, C.POINTER(SearchParams), vp, i64, ip, vp, C.POINTER(i64), ip, C.c_int, vp, ip, C.POINTER(i64),
                                               vp, vp, C.c_int, C.c_int, C.POINTER(i64), vp, vp, C.c_int, C.POINTER(PictureStats)]
        L.havoc_search_picture_uni.restype = C.c_int
        L.havoc_search_picture_uni_device.argtypes = [vp, C.c_int, C.POINTER(SearchParams), vp, i64, ip, vp, C.POINTER(i64), ip, C.c_int, vp, ip, C.POINTER(i64),
                                                      vp, vp, C.c_int, C.c_int, C.POINTER(i64), vp, vp, vp, vp, C.POINTER(PictureStats)]
        L.havoc_search_picture_uni_device.restype = C.c_int
        L.havoc_search_rqt.argtypes = [vp, C.c_int, C.c_int, vp, i64, ip, vp, ip, vp, i64, ip, vp, vp, C.c_double, C.c_double, C.c_int, vp, C.c_int, vp,
                                       C.POINTER(RqtStats)]
        L.havoc_search_rqt.restype = C.c_int
        L.havoc_search_intra_rd.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, ip, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp,
                                            C.POINTER(RqtStats)]
        L.havoc_search_intra_rd.restype = C.c_int
        L.havoc_search_intra_device.argtypes = [vp, C.c_int, C.c_int, vp, ip, vp, C.c_int, vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(RqtStats)]
        L.havoc_search_intra_device.restype = C.c_int
        L.havoc_search_intra_modes.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, ip, vp, vp, C.c_int, vp, C.c_double, vp, vp]
        L.havoc_search_intra_modes.restype = C.c_int
        L.havoc_search_intra_chain.argtypes = [vp, C.c_int, C.c_int, vp, vp, ip, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, vp]
        L.havoc_search_intra_chain.restype = C.c_int
        L.havoc_search_block_cells.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        L.havoc_search_block_cells.restype = C.c_int
        L.havoc_search_release.argtypes = [vp]
        L.havoc_search_release.restype = None
        _lib = L
    return _lib


def picture_uni(ctx, S, params, d_src, src_origin, src_stride, d_ref, ref_origin, ref_stride, ref_pad, d_phase, plane_elems, phase_origin, pus, ctu_first,
                ctus_x, ctus_y, mvp_rate=(65536, 65536), threads=16, want_field=True, on_device=False, d_field_keep=None, bi=False):
    """havoc_search_picture_uni (launch + host replay rounds) or, on_device, havoc_search_picture_uni_device (the decision loops inside the kernel).
    ctx: havoc_mi355x context handle (int / c_void_p); d_*: device addresses (ints); ref_origin / phase_origin:
    pairs (list 0, list 1).  Returns (results [2 * len(pus)] RESULT_DT indexed 2 * p + list, field int16 [2, cells_y, cells_x, 2] or None, stats);
    with bi (device search only) the bi-directional refinements [2 * len(pus)] as a fourth value."""
    pus = np.ascontiguousarray(pus)
    assert pus.dtype == PICTURE_PU_DT
    ctu_first = np.ascontiguousarray(ctu_first, np.int32)
    out = np.zeros(2 * len(pus), RESULT_DT)
    cw, ch = (params.pic_width + 3) // 4, (params.pic_height + 3) // 4
    field = np.zeros((2, ch, cw, 2), np.int16) if want_field else None
    stats = PictureStats()
    ro = (C.c_int64 * 2)(*[int(v) for v in ref_origin])
    po = (C.c_int64 * 2)(*[int(v) for v in phase_origin])
    mr = (C.c_int64 * 2)(*[int(v) for v in mvp_rate])
    out_bi = np.zeros(2 * len(pus), RESULT_DT) if bi else None
    if bi and not on_device:
        raise ValueError("the bi-directional refinement inside the picture walk exists in the device search only")
    if on_device:
        rc = lib().havoc_search_picture_uni_device(ctx, S, C.byref(params), d_src, int(src_origin), src_stride, d_ref, ro, ref_stride, ref_pad, d_phase, plane_elems, po,
                                                   pus.ctypes.data, ctu_first.ctypes.data, ctus_x, ctus_y, mr, out.ctypes.data,
                                                   field.ctypes.data if want_field else None, d_field_keep, out_bi.ctypes.data if bi else None, C.byref(stats))
    else:
        rc = lib().havoc_search_picture_uni(ctx, S, C.byref(params), d_src, int(src_origin), src_stride, d_ref, ro, ref_stride, ref_pad, d_phase, plane_elems, po,
                                            pus.ctypes.data, ctu_first.ctypes.data, ctus_x, ctus_y, mr, out.ctypes.data,
                                            field.ctypes.data if want_field else None, threads, C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"havoc_search_picture_uni{'_device' if on_device else ''} failed ({rc})")
    return (out, field, stats, out_bi) if bi else (out, field, stats)


def decision_inputs(width, height, bit_depth=8, qp=32, seed=11, density=1.0, frames=None, distance=1):
    """host side of a DecisionPicture: padded luma planes (source, list 0, list 1), the picture's PUs in decision order, search parameters.
    distance: temporal distance of the two reference pictures (1 = the leaf B pictures of the hierarchy, half of a SOP of 8; 2, 4, 8 = its upper
    layers, whose vectors are longer and whose searches run the star / raster refinement far more often)"""
    from . import workload
    pad = 96
    if frames is None:
        frames = workload.synth_frames(width, height, 2 * distance + 1, seed, bit_depth)
        frames = [frames[0], frames[distance], frames[2 * distance]]
    planes = [workload.pad_plane(f[0], pad) for f in (frames[1], frames[0], frames[2])]
    stride = planes[0].shape[1]
    # chroma: Cb of (source, list 0, list 1), then Cr, in the same padded layout at half size (48 samples of border)
    chroma = [workload.pad_plane(f[c], pad // 2) for c in (1, 2) for f in (frames[1], frames[0], frames[2])]
    pus, ctu_first, cx, cy = workload.picture_pus(width, height, seed, density)
    lam = workload.picture_lambda(qp)
    return dict(chroma=[np.ascontiguousarray(p.ravel()) for p in chroma], cstride=chroma[0].shape[1],
                planes=[np.ascontiguousarray(p.ravel()) for p in planes], stride=stride, pad=pad, pus=pus, ctu_first=ctu_first, cx=cx, cy=cy,
                params=medium_params(width, height, bit_depth, 1.0 / np.sqrt(lam)), mvp_rate=(45000, 98000), lam=lam)


def rqt(ctx, S, bit_depth, d_src, src_origin, src_stride, d_pred, pred_stride, d_rec, rec_origin, rec_stride, d_states, quant, lam, reciprocal_lambda, cus, sdh=1):
    """havoc_search_rqt: the residual-quadtree decision of every inter unit of `cus` (RQT_CU_DT) in one chain per transform size.
    Returns (results RQT_RESULT_DT, stats)"""
    cus = np.ascontiguousarray(cus)
    assert cus.dtype == RQT_CU_DT
    quant = np.ascontiguousarray(quant, np.int32)
    out = np.zeros(len(cus), RQT_RESULT_DT)
    stats = RqtStats()
    rc = lib().havoc_search_rqt(ctx, S, bit_depth, d_src, int(src_origin), src_stride, d_pred, pred_stride, d_rec, int(rec_origin), rec_stride, d_states,
                                quant.ctypes.data, float(lam), float(reciprocal_lambda), int(sdh), cus.ctypes.data, len(cus), out.ctypes.data, C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"havoc_search_rqt failed ({rc})")
    return out, stats


def intra_rd(ctx, S, bit_depth, log2, d_src, src_stride, d_neighbours, jobs, order, ictx, ctx_index, d_states, quant_row, lam, reciprocal_lambda, d_rec, sdh=1):
    """havoc_search_intra_rd: the RD refinement of n intra partitions of one size (jobs: int32 [n, 8] havoc_mi355x_intra_search_job rows on the
    HOST; order: INTRA_RESULT_DT from the 35-mode stage; quant_row: int32[4] of rqt_quant for this size).  Returns (INTRA_RD_RESULT_DT[n], stats)"""
    jobs = np.ascontiguousarray(jobs, np.int32)
    order = np.ascontiguousarray(order)
    ictx = np.ascontiguousarray(ictx)
    ctx_index = np.ascontiguousarray(ctx_index, np.int32)
    quant_row = np.ascontiguousarray(quant_row, np.int32)
    out = np.zeros(len(jobs), INTRA_RD_RESULT_DT)
    stats = RqtStats()
    rc = lib().havoc_search_intra_rd(ctx, S, bit_depth, log2, d_src, src_stride, d_neighbours, jobs.ctypes.data, len(jobs), order.ctypes.data, ictx.ctypes.data,
                                     ctx_index.ctypes.data, d_states, quant_row.ctypes.data, float(lam), float(reciprocal_lambda), int(sdh), d_rec, out.ctypes.data,
                                     C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"havoc_search_intra_rd failed ({rc})")
    return out, stats


INTRA_GROUP_DT = np.dtype([("log2", "i4"), ("n", "i4"), ("d_neighbours", "u8"), ("d_jobs", "u8"), ("d_ictx", "u8"), ("d_ctx_index", "u8"), ("d_rec", "u8"),
                           ("out", "u8")])                                                                       # havoc_intra_group
assert INTRA_GROUP_DT.itemsize == 56


INTRA_CHAIN_SIZE_DT = np.dtype([("log2", "i4"), ("n", "i4"), ("d_neighbours", "u8"), ("d_jobs", "u8"), ("d_ictx", "u8"), ("d_ctx_index", "u8"), ("d_parts", "u8"),
                                ("d_blocks", "u8"), ("first", "u8"), ("out", "u8")])                                  # havoc_intra_chain_size
assert INTRA_CHAIN_SIZE_DT.itemsize == 72


def _address(p):
    return int(p.value or 0) if isinstance(p, C.c_void_p) else int(p)


def intra_device(ctx, S, bit_depth, d_src, src_stride, groups, d_states, quant, reciprocal_sqrt_lambda, lam, reciprocal_lambda, sdh=1):
    """havoc_search_intra_device: both stages of the intra partitions of all sizes, the decisions between the launches taken on the device.
    groups: dicts(log2, n, d_nb, d_jobs, d_ictx, d_ctu, d_rec) of device addresses; quant: int32 [4, 4] (rqt_quant).
    Returns ({log2: INTRA_RD_RESULT_DT[n]}, stats)"""
    table = np.zeros(len(groups), INTRA_GROUP_DT)
    outs = {}
    for row, g in zip(table, groups):
        out = np.zeros(g["n"], INTRA_RD_RESULT_DT)
        outs[g["log2"]] = out
        row["log2"], row["n"] = g["log2"], g["n"]
        for k, name in (("d_nb", "d_neighbours"), ("d_jobs", "d_jobs"), ("d_ictx", "d_ictx"), ("d_ctu", "d_ctx_index"), ("d_rec", "d_rec")):
            row[name] = _address(g[k])
        row["out"] = out.ctypes.data
    quant = np.ascontiguousarray(quant, np.int32)
    stats = RqtStats()
    rc = lib().havoc_search_intra_device(ctx, S, bit_depth, d_src, src_stride, table.ctypes.data, len(table), d_states, quant.ctypes.data, float(reciprocal_sqrt_lambda),
                                         float(lam), float(reciprocal_lambda), int(sdh), C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"havoc_search_intra_device failed ({rc})")
    return outs, stats


def intra_modes(ctx, S, bit_depth, log2, d_src, src_stride, d_neighbours, d_jobs, n, ictx, reciprocal_sqrt_lambda):
    """havoc_search_intra_modes: the 35-mode SATD stage of n partitions of one size + the order their modes are refined in (Search.hpp:40-190).
    d_jobs: device copy of the int32 [n, 8] job rows.  Returns INTRA_RESULT_DT[n]"""
    ictx = np.ascontiguousarray(ictx)
    out = np.zeros(n, INTRA_RESULT_DT)
    rc = lib().havoc_search_intra_modes(ctx, S, bit_depth, log2, d_src, src_stride, d_neighbours, d_jobs, n, ictx.ctypes.data, float(reciprocal_sqrt_lambda),
                                        out.ctypes.data, None)
    if rc != 0:
        raise RuntimeError(f"havoc_search_intra_modes failed ({rc})")
    return out


def intra_chain_tables(width, height, bit_depth, qp, seed, stride, pad):
    """host side of an intra picture's dependency chain (IntraChainPicture; tests/intra_chain_runner.py drives the same tables through the CPU stand-in device): the
    partitions, their owner map and levels (workload.intra_picture_partitions), and per partition size -- ordered by level -- the job records, the most-probable-mode
    records (rates and max_refine; the modes are derived from the neighbours' champions as the chain runs), CABAC snapshot indices, chain records and level slices"""
    from . import workload
    parts, owner, level = workload.intra_picture_partitions(width, height, seed + 3)
    cx = (width + 63) // 64
    lam = workload.picture_lambda(qp)
    rng = np.random.default_rng(seed + 7919)
    base = rng.integers(4, 100, 128)
    rdoq_states = np.clip(base[None, :] + rng.integers(-6, 7, (cx * ((height + 63) // 64), 128)), 0, 125).astype(np.uint8)
    nlevels = int(level.max()) + 1
    sizes = {}
    for log2 in (5, 4, 3, 2):
        sel = np.flatnonzero(parts["log2"] == log2)
        if not len(sel):
            continue
        sel = sel[np.argsort(level[sel], kind="stable")]
        m, nn = len(sel), 1 << log2
        L = 4 * nn + 1
        x0, y0 = parts["x0"][sel].astype(np.int64), parts["y0"][sel].astype(np.int64)
        mask = workload.intra_filter_mask(nn)
        jobs = np.zeros((m, 8), np.int64)
        jobs[:, 0] = (y0 + pad) * stride + x0 + pad
        jobs[:, 1] = np.arange(m) * 2 * L + 2 * nn + 1
        jobs[:, 2] = jobs[:, 1] + L
        jobs[:, 3], jobs[:, 4], jobs[:, 5] = mask & 0xffffffff, mask >> 32, 1
        jobs = jobs.astype(np.uint32).view(np.int32).reshape(m, 8)
        ictx = np.zeros(m, INTRA_CTX_DT)
        ictx["max_refine"] = 3 if log2 > 3 else 8            # Speed::nCandidatesIntraRefinement at medium
        ictx["rate_a_minus_c"] = -rng.integers(300000, 420000, m)
        ictx["rate_b_minus_c"] = -rng.integers(100000, 200000, m)
        chain = np.zeros((m, 4), np.int32)
        chain[:, 0], chain[:, 1], chain[:, 2], chain[:, 3] = x0, y0, log2, sel
        ctu = ((y0 // 64) * cx + x0 // 64).astype(np.int32)
        first = np.searchsorted(level[sel], np.arange(nlevels + 1)).astype(np.int64)
        sizes[log2] = dict(sel=sel, m=m, nn=nn, jobs=jobs, ictx=ictx, ctu=ctu, chain=chain, first=first)
    return dict(parts=parts, owner=owner, level=level, nlevels=nlevels, cx=cx, lam=lam, rsl=float(1.0 / np.sqrt(lam)), quant=rqt_quant(qp, bit_depth), rdoq_states=rdoq_states,
                sizes=sizes)


class IntraChainPicture:
    """An INTRA picture with the real dependencies between its partitions (round 4): every partition predicts from the reconstruction
    of the partitions before it and takes its most probable modes from its neighbours' champions (turing/Reconstruct.cpp:609-615, CandModeList.h:33-95), so
    the picture is a dependency graph, cut here into LEVELS of mutually independent partitions (workload.intra_picture_partitions).  A level is one pass of the
    batch chain: reference samples + candModeList gathered on the device from the running reconstruction / mode map (havoc_mi355x_intra_gather), the 35-mode
    SATD stage, the refinement order, every candidate through prediction -> T -> RDOQ -> IQ -> IT -> SSD, the champion (havoc_search_intra_device: decisions on the
    device), and the champions' blocks + modes committed to the picture (havoc_mi355x_intra_commit).  Levels are many (a 4x4 partition's chain reaches across the
    picture: ~600 at 640x360, ~1 900 at 1080p) and each is launch-bound -- the whole picture inside one persistent kernel is the form that would pay; this one is
    the parity-checked statement of the chain."""

    PAD = 96

    def __init__(self, hv, width, height, bit_depth=8, qp=32, seed=11):
        import torch
        from . import havoc as hmod
        from . import workload
        self.hv, self.torch = hv, torch
        self.W, self.H, self.bd, self.qp = width, height, bit_depth, qp
        self.S = 1 if bit_depth == 8 else 2
        self.dt = np.uint8 if self.S == 1 else np.uint16
        src = workload.pad_plane(workload.synth_frames(width, height, 1, seed, bit_depth)[0][0], self.PAD)
        self.stride = src.shape[1]
        self.host_src = np.ascontiguousarray(src.ravel())
        self.n = self.host_src.size
        self.d_src = hv.up(self.host_src)
        self.d_rec = hv.zeros(self.n, self.dt)
        t = intra_chain_tables(width, height, bit_depth, qp, seed, self.stride, self.PAD)
        self.parts, self.owner, self.level, self.nlevels = t["parts"], t["owner"], t["level"], t["nlevels"]
        self.cells_per_row = self.owner.shape[1]
        self.d_owner = hv.up(np.ascontiguousarray(self.owner.ravel()))
        self.d_modes = hv.zeros(self.owner.size, np.uint8)
        self.cx, self.lam, self.rsl, self.quant, self.rdoq_states = t["cx"], t["lam"], t["rsl"], t["quant"], t["rdoq_states"]
        self.d_states = hv.up(self.rdoq_states.reshape(-1))
        self.layout = hv.intra_chain_layout(width, height, self.stride, self.PAD, self.cells_per_row, bit_depth)
        self.sizes = {}
        for log2, g in t["sizes"].items():
            m, nn = g["m"], g["nn"]
            self.sizes[log2] = dict(g, d_jobs=hv.up(g["jobs"]), d_nb=hv.zeros(m * 2 * (4 * nn + 1), self.dt), d_ictx=hv.up(np.ascontiguousarray(g["ictx"]).view(np.int32)),
                                    d_ctu=hv.up(g["ctu"]), d_parts=hv.up(g["chain"]), d_blocks=hv.zeros(m * nn * nn, self.dt), d_mode=hv.zeros(m, np.int32),
                                    best=np.zeros(m, INTRA_RD_RESULT_DT))
        hv.sync()

    def step(self, wait_per_level=False):
        """the picture, level by level; leaves the champions in self.sizes[log2]["best"] (ordered as ["sel"]), the reconstruction in self.d_rec.
        Default: havoc_search_intra_chain -- every level's launches queued without a wait between the levels (the chain runs over the worst-case number of
        candidate slots), one wait at the end.  wait_per_level: the first form of this step, a havoc_search_intra_device call (two waits) per level."""
        hv, S, torch = self.hv, self.S, self.torch
        self.launches = 0
        if not wait_per_level:
            table = np.zeros(len(self.sizes), INTRA_CHAIN_SIZE_DT)
            firsts = []
            for row, (log2, g) in zip(table, self.sizes.items()):
                f = np.ascontiguousarray(g["first"], np.int32)
                firsts.append(f)
                row["log2"], row["n"] = log2, g["m"]
                row["d_neighbours"], row["d_jobs"], row["d_ictx"], row["d_ctx_index"] = g["d_nb"].data_ptr(), g["d_jobs"].data_ptr(), g["d_ictx"].data_ptr(), g["d_ctu"].data_ptr()
                row["d_parts"], row["d_blocks"], row["first"], row["out"] = g["d_parts"].data_ptr(), g["d_blocks"].data_ptr(), f.ctypes.data, g["best"].ctypes.data
            quant = np.ascontiguousarray(self.quant, np.int32)
            stats = RqtStats()
            rc = lib().havoc_search_intra_chain(hv.h, S, self.bd, self.layout, self.d_src.data_ptr(), self.stride, self.d_rec.data_ptr(), self.d_owner.data_ptr(),
                                                self.d_modes.data_ptr(), table.ctypes.data, len(table), self.nlevels, self.d_states.data_ptr(), quant.ctypes.data, self.rsl,
                                                float(self.lam), 1.0 / self.lam, 1, C.byref(stats))
            if rc != 0:
                raise RuntimeError(f"havoc_search_intra_chain failed ({rc})")
            self.launches, self.chain_stats = stats.launches, stats
            return
        for lvl in range(self.nlevels):
            groups, live = [], []
            for log2, g in self.sizes.items():
                a, b = int(g["first"][lvl]), int(g["first"][lvl + 1])
                if b == a:
                    continue
                n, area = b - a, g["nn"] * g["nn"]
                hv.intra_gather_a(S, self.layout, self.d_rec.data_ptr(), self.d_owner.data_ptr(), self.d_modes.data_ptr(), g["d_parts"].data_ptr() + 16 * a, n,
                                  g["d_jobs"].data_ptr() + 32 * a, g["d_nb"].data_ptr(), g["d_ictx"].data_ptr() + 40 * a)
                groups.append(dict(log2=log2, n=n, d_nb=g["d_nb"].data_ptr(), d_jobs=g["d_jobs"].data_ptr() + 32 * a, d_ictx=g["d_ictx"].data_ptr() + 40 * a,
                                   d_ctu=g["d_ctu"].data_ptr() + 4 * a, d_rec=g["d_blocks"].data_ptr() + a * area * S))
                live.append((log2, g, a, b))
            best, st = intra_device(hv.h, S, self.bd, self.d_src.data_ptr(), self.stride, groups, self.d_states.data_ptr(), self.quant, self.rsl, self.lam, 1.0 / self.lam)
            self.launches += st.launches + 2 * len(live)
            for log2, g, a, b in live:
                g["best"][a:b] = best[log2]
                with torch.cuda.stream(hv.tstream):
                    g["d_mode"][a:b].copy_(torch.from_numpy(np.ascontiguousarray(best[log2]["mode"])), non_blocking=False)
                hv.intra_commit_a(S, self.layout, self.d_rec.data_ptr(), self.d_modes.data_ptr(), g["d_parts"].data_ptr() + 16 * a, b - a,
                                  g["d_blocks"].data_ptr() + a * g["nn"] * g["nn"] * S, g["d_mode"].data_ptr() + 4 * a)
        hv.sync()

    def results(self):
        """(champions INTRA_RD_RESULT_DT [partitions] in CODING order, candModeList the device derived [partitions, 3], neighbour_modes, reconstruction plane)"""
        hv = self.hv
        best = np.zeros(len(self.parts), INTRA_RD_RESULT_DT)
        cand, nbm = np.zeros((len(self.parts), 3), np.int32), np.zeros(len(self.parts), np.int32)
        for g in self.sizes.values():
            best[g["sel"]] = g["best"]
            ic = hv.down(g["d_ictx"], np.int32).view(INTRA_CTX_DT)
            cand[g["sel"]], nbm[g["sel"]] = ic["cand_mode_list"], ic["neighbour_modes"]
        return best, cand, nbm, hv.down(self.d_rec, self.dt)


def rqt_units(width, height, ctus_x):
    """the inter units whose transform trees are decided: 32x32 units where they fit, 16x16 then 8x8 units along a partial last row / column"""
    rows = []
    for size, log2 in ((32, 5), (16, 4), (8, 3)):
        for y in range(0, height - size + 1, size):
            for x in range(0, width - size + 1, size):
                # a unit is emitted at the largest size that fits its position on the grid of that size and lies outside the area covered by larger units
                big = size * 2
                covered = size < 32 and x // big * big + big <= width and y // big * big + big <= height
                if not covered:
                    rows.append((x, y, log2, (y // 64) * ctus_x + x // 64))
    return np.array(rows, np.int32).reshape(-1, 4).view(RQT_CU_DT).reshape(-1)


class DecisionPicture:
    """One inter picture through the DECISION-DRIVEN path on the device (bench.py --decisions, tests/test_decisions.py):

      1. the 15 fractional-sample planes of both reference pictures (havoc_mi355x_interp_planes);
      2. every PU's uni-directional search in both lists, CTUs in wavefront order, predictors derived from the vectors decided before, then the
         bi-directional refinement of every PU (searchBi) -- the decision loops run inside the kernel (libhavoc_search.so:
         havoc_search_picture_uni_device -> csrc/kernels_search.hip; search_on_device=False: launch + host replay rounds, no bi refinement);
      3. the TU side ON THE CHOSEN VECTORS: HavocPredUni of every 16x16 block (8x8 in a last partial row) at the list-0 vector the
         search left for it, then the residual-quadtree decision of every inter unit (32x32 units, smaller along partial edges;
         libhavoc_search.so: havoc_search_rqt): both tree depths of every unit through residual + forward DCT -> Rdoq::runQuantisation ->
         de-quantise + inverse DCT + add -> SSD in one chain per transform size, the decisions taken from 16 bytes per candidate, the chosen
         candidates reconstructed into the picture (job tables are built from the decided motion field: they cannot exist before 2).

      4. the intra candidates an inter picture evaluates (21.4 k partitions per 1080p frame): per partition size the 35-mode
         prediction + SATD stage and its refinement order (havoc_search_intra_modes), then every candidate mode reconstructed and the champion
         picked (havoc_search_intra_rd) -- with neighbours from the SOURCE picture, i.e. without the chain through the previous partition's
         reconstruction that the encoder has (stated; that chain needs a device-side loop).

    What is NOT in it (stated, not hidden): the encoder's mode decision between the searched PUs (every PU of workload.picture_pus is
    searched and the last one covering an area stands) and between inter and intra, bi-prediction, CABAC.  `step()` is what bench.py times."""

    PAD = 96

    def __init__(self, hv, width, height, bit_depth=8, qp=32, seed=11, threads=16, frames=None, density=1.0, intra=True, search_on_device=True, distance=1):
        import torch
        from . import havoc as hmod
        from . import workload
        self.hv, self.torch, self.hmod = hv, torch, hmod
        self.W, self.H, self.bd, self.qp, self.threads = width, height, bit_depth, qp, threads
        self.search_on_device = search_on_device      # the decision loops inside the kernel (kernels_search.hip) / launch + host replay rounds
        self.use_graphs, self._graphs = True, {}
        self.S = 1 if bit_depth == 8 else 2
        self.dt = np.uint8 if self.S == 1 else np.uint16
        d = decision_inputs(width, height, bit_depth, qp, seed, density, frames, distance)
        self.stride = d["stride"]
        self.host_planes = d["planes"]                            # source, list 0, list 1
        self.n = self.host_planes[0].size
        self.pe = (self.n + 63) & ~63
        self.origin = self.PAD * self.stride + self.PAD
        pic = np.zeros(3 * self.pe, self.dt)
        for k, p in enumerate(self.host_planes):
            pic[k * self.pe:k * self.pe + self.n] = p
        self.d_pic = hv.up(pic)                                   # one allocation: a job names a plane by a 32-bit sample offset
        self.d_phase = hv.zeros(32 * self.pe, self.dt)            # 2 references x 16 phase planes
        self.pus, self.ctu_first, self.cx, self.cy = d["pus"], d["ctu_first"], d["cx"], d["cy"]
        lam = d["lam"]
        self.params, self.mvp_rate = d["params"], d["mvp_rate"]
        # ---- prediction: 16x16 blocks over the rows that hold whole ones, 8x8 blocks over a last partial row; the (older) fixed-size TU chain
        # over the same blocks stays available as tu_chain_fixed()
        self.units = rqt_units(width, height, self.cx)
        self.quant = rqt_quant(qp, bit_depth)
        self.lam = lam
        self.groups = []
        h16 = height // 16 * 16
        for log2, y_lo, y_hi in ((4, 0, h16), (3, h16, height // 8 * 8)):
            nn = 1 << log2
            if y_hi <= y_lo:
                continue
            xs, ys = np.meshgrid(np.arange(0, width // nn * nn, nn), np.arange(y_lo, y_hi, nn))
            x0, y0 = xs.ravel().astype(np.int64), ys.ravel().astype(np.int64)
            m = len(x0)
            qs, qshift, _ = workload.quant_params(qp, log2, bit_depth, False)
            inv, dshift = workload.dequant_params(qp, log2, bit_depth)
            fj = np.zeros((m, 4), np.int32)
            fj[:, 0] = np.arange(m) * nn * nn
            fj[:, 1] = fj[:, 3] = (y0 + self.PAD) * self.stride + x0 + self.PAD
            fj[:, 2] = y0 * width + x0
            jobs = np.zeros(m, hmod.RDOQ_JOB_DT)
            jobs["dst_off"] = jobs["src_off"] = fj[:, 0]
            jobs["quant_scale"], jobs["quant_shift"], jobs["inv_scale"] = qs, qshift, inv
            jobs["lambda_q16"], jobs["sdh_factor"] = hmod.rdoq_lambda(lam, inv)
            jobs["sdh"] = 1
            jobs["ctx_index"] = (y0 // 64) * self.cx + x0 // 64
            with torch.cuda.stream(hv.tstream):
                d_rj = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(hv.device)
            self.groups.append(dict(log2=log2, nn=nn, x0=x0, y0=y0, m=m, inv=inv, dshift=dshift, d_fj=hv.up(fj), d_rj=d_rj,
                                    pj=np.zeros((m, 8), np.int32), d_pj=hv.zeros(m * 8, np.int32), coef=hv.zeros(m * nn * nn, np.int16),
                                    level=hv.zeros(m * nn * nn, np.int16), cbf=hv.zeros(m, np.int32), ssd=hv.zeros(m, np.uint32),
                                    work=hv.rdoq_workspace(m)))
        srng = np.random.default_rng(seed + 7919)
        base = srng.integers(4, 100, 128)
        self.rdoq_states = np.clip(base[None, :] + srng.integers(-6, 7, (self.cx * self.cy, 128)), 0, 125).astype(np.uint8)
        self.d_states = hv.up(self.rdoq_states.reshape(-1))
        self.pred = hv.zeros(width * height, self.dt)
        self.recon = hv.zeros(self.pe, self.dt)
        # chroma (round 4): Cb / Cr of source, list 0, list 1 in ONE allocation (plane k at k * cpe: 0-2 Cb, 3-5 Cr), their prediction and reconstruction planes
        self.host_chroma, self.cstride = d["chroma"], d["cstride"]
        self.cn = self.host_chroma[0].size
        self.cpe = (self.cn + 63) & ~63
        self.corigin = (self.PAD // 2) * self.cstride + self.PAD // 2
        cpic = np.zeros(6 * self.cpe, self.dt)
        for k, p in enumerate(self.host_chroma):
            cpic[k * self.cpe:k * self.cpe + self.cn] = p
        self.d_cpic = hv.up(cpic)
        self.cpred = hv.zeros(2 * (width // 2) * (height // 2), self.dt)      # Cb then Cr, unpadded, stride width / 2
        # the decided motion field stays on the device (int16 [2 lists][cells][2]); the job tables of the steps after the searches are made from it THERE
        self.d_field = hv.zeros(2 * ((width + 3) // 4) * ((height + 3) // 4) * 2, np.int16)
        self.layout = hv.field_layout(width, height, self.stride, self.PAD, self.pe, self.cstride, self.PAD // 2, self.cpe)
        self.crecon = hv.zeros(2 * self.cpe, self.dt)
        # ---- the intra candidates of the picture (an inter picture evaluates them per coding unit): partitions with neighbours from the source
        self.intra_parts = {}
        self.rsl = float(self.params.reciprocal_sqrt_lambda)
        if intra:
            src2d = self.host_planes[0].reshape(-1, self.stride)
            for log2, (jobs, nb, ictx, ctu) in workload.intra_partitions(src2d, width, height, self.PAD, seed + 31).items():
                self.intra_parts[log2] = dict(jobs=jobs, nb=nb, ictx=ictx, ctu=ctu, d_jobs=hv.up(jobs), d_nb=hv.up(nb),
                                              d_ictx=hv.up(np.ascontiguousarray(ictx).view(np.int32)), d_ctu=hv.up(np.ascontiguousarray(ctu, np.int32)),
                                              d_rec=hv.zeros(len(jobs) << (2 * log2), self.dt))
        hv.sync()

    def phase_planes(self):
        hv, pe, S = self.hv, self.pe, self.S
        for r in (0, 1):
            ph = self.d_phase[r * 16 * pe:(r + 1) * 16 * pe]
            ref = self.d_pic[(1 + r) * pe:(2 + r) * pe]
            with self.torch.cuda.stream(hv.tstream):
                ph[:pe] = ref                                   # phase 0 = the picture itself
            hv.interp_planes_d(self.bd, ph, pe, ref, self.stride, 12, 4, self.W + 2 * self.PAD - 24, self.H + 2 * self.PAD - 8)

    def search(self):
        hv, pe, o = self.hv, self.pe, self.origin
        base, ph = self.d_pic.data_ptr(), self.d_phase.data_ptr()
        r = picture_uni(hv.h, self.S, self.params, base, o, self.stride, base, (pe + o, 2 * pe + o), self.stride, self.PAD, ph, pe, (o, 16 * pe + o),
                        self.pus, self.ctu_first, self.cx, self.cy, self.mvp_rate, self.threads, on_device=self.search_on_device, bi=self.search_on_device,
                        d_field_keep=self.d_field.data_ptr() if self.search_on_device else None)
        self.bi_results = r[3] if self.search_on_device else None      # the bi-directional refinements (device search only)
        if not self.search_on_device:                                  # the host-replay search decides on the host: its field goes down once
            with self.torch.cuda.stream(hv.tstream):
                self.d_field.copy_(self.torch.from_numpy(r[1].reshape(-1)), non_blocking=False)
        return r[:3]

    def predict(self, field):
        """HavocPredUni of every inter unit (a 2Nx2N prediction unit per unit of rqt_units) at the list-0 vector decided at its origin, into the
        prediction plane; asynchronous"""
        hv, bd = self.hv, self.bd
        if not hasattr(self, "pgroups"):
            self.pgroups = []
            for log2 in (5, 4, 3):
                sel = np.flatnonzero(self.units["log2_size"] == log2)
                if len(sel):
                    x0, y0 = self.units["x0"][sel].astype(np.int32), self.units["y0"][sel].astype(np.int32)
                    self.pgroups.append(dict(log2=log2, nn=1 << log2, d_x0=hv.up(x0), d_y0=hv.up(y0), d_dst=hv.up(y0 * self.W + x0), d_pj=hv.zeros(len(sel) * 8, np.int32)))
        for g in self.pgroups:
            # the job table from the field, on the device (k_pred_jobs): reference offsets count from the start of the luma allocation
            hv.pred_jobs_d(self.layout, self.d_field, 0, g["d_x0"], g["d_y0"], g["log2"], 0, g["d_dst"], g["d_pj"])
            hv.pred_uni_d(8, bd, self.pred, self.W, self.d_pic, self.stride, g["d_pj"].view(-1, 8), g["nn"], g["nn"])

    # ---- round 4: the merge candidates of every unit and the chroma planes -------------------------------------------------
    MERGE_CANDIDATES = 5

    def merge_vectors(self, field):
        """the candidate vectors of every unit of self.units: both lists' vectors decided for the cells at the five spatial merge positions (HEVC 8.5.3.2.3:
        A1 left-bottom, B1 above-right, B0, A0, B2 -- Mvp.h's derivation with its pruning and temporal candidate stays out of scope: the vectors are INPUTS);
        a position outside the picture gives zero vectors; every vector is limited so that the block and its filter taps stay inside the padded planes
        (as LimitFullPelMv does for the searches).  int16 [units, 5, 2 lists, 2]"""
        u = self.units
        n = 1 << u["log2_size"].astype(np.int64)
        x0, y0 = u["x0"].astype(np.int64), u["y0"].astype(np.int64)
        px = np.stack([x0 - 1, x0 + n - 1, x0 + n, x0 - 1, x0 - 1], 1)
        py = np.stack([y0 + n - 1, y0 - 1, y0 - 1, y0 + n, y0 - 1], 1)
        inside = (px >= 0) & (py >= 0) & (px < self.W) & (py < self.H)
        cx, cy = np.clip(px, 0, self.W - 1) >> 2, np.clip(py, 0, self.H - 1) >> 2
        mv = np.stack([field[0, cy, cx], field[1, cy, cx]], 2).astype(np.int64)      # [units, 5, list, xy]
        mv *= inside[:, :, None, None]
        lo_x, hi_x = (-64 - x0) * 4, (self.W + 64 - x0 - n) * 4
        lo_y, hi_y = (-64 - y0) * 4, (self.H + 64 - y0 - n) * 4
        mv[..., 0] = np.clip(mv[..., 0], lo_x[:, None, None], hi_x[:, None, None])
        mv[..., 1] = np.clip(mv[..., 1], lo_y[:, None, None], hi_y[:, None, None])
        return mv.astype(np.int16)

    def merge_candidates(self, field):
        """searchMergeModes / measurePuCost (turing/Search.hpp:1659-1706, 1754-1768) as batches: every unit's five candidates predicted bi-directionally in all
        three planes (HavocPredBi 8-tap luma, 4-tap Cb / Cr) and measured with the Hadamard SATD against the source (Measure.h:97-168: chroma only where the
        halved unit is a multiple of 4), cost = rate + (satdY + satdCb + satdCr) * reciprocalSqrtLambda with a stand-in rate of the merge index (i + 1 bits,
        4 at most).  Leaves self.merge = dict(vectors, satd [units, 5, 3], cost [units, 5] Q16, best [units]).  9 launches per unit size."""
        hv, bd, torch = self.hv, self.bd, self.torch
        u, K = self.units, self.MERGE_CANDIDATES
        if not hasattr(self, "mgroups"):
            # everything that does not depend on the vectors is made once: the units' positions, destination slots, the source blocks' SATD jobs
            self.mgroups, at = [], 0
            for log2 in (5, 4, 3):
                sel = np.flatnonzero(u["log2_size"] == log2)
                if not len(sel):
                    continue
                nn, m = 1 << log2, len(sel) * K
                x0, y0 = np.repeat(u["x0"][sel].astype(np.int64), K), np.repeat(u["y0"][sel].astype(np.int64), K)
                g = dict(log2=log2, nn=nn, sel=sel, m=m, planes=[], d_x0=hv.up(u["x0"][sel].astype(np.int32)), d_y0=hv.up(u["y0"][sel].astype(np.int32)),
                         d_vec=hv.zeros(m * 4, np.int16), d_cost=torch.zeros(m, dtype=torch.int64, device=hv.device), d_best=hv.zeros(len(sel), np.int32))
                for plane in range(3):
                    c = plane > 0
                    size, stride, pad, pe = (nn // 2, self.cstride, self.PAD // 2, self.cpe) if c else (nn, self.stride, self.PAD, self.pe)
                    bx, by = (x0 // 2, y0 // 2) if c else (x0, y0)
                    first = (3 * (plane - 1)) if c else 0                # plane index of the source inside the allocation (Cb: 0, Cr: 3; luma: 0)
                    dst = np.arange(m) * size * size
                    sj = np.stack([first * pe + (by + pad) * stride + bx + pad, dst, np.full(m, size), np.full(m, size)], 1).astype(np.int32)
                    g["planes"].append(dict(size=size, stride=stride, taps=4 if c else 8, d_bj=hv.zeros(m * 12, np.int32), d_sj=hv.up(sj),
                                            d_dst=hv.zeros(m * size * size, self.dt), at=at))
                    at += m
                self.mgroups.append(g)
            self.d_merge_satd = hv.zeros(at, np.int32)      # every SATD of the step in one buffer
        lam_q16 = int(float(self.params.reciprocal_sqrt_lambda) * 65536 + 0.5)
        for g in self.mgroups:
            q = g["planes"]
            hv.merge_jobs_d(self.layout, self.d_field, g["d_x0"], g["d_y0"], g["log2"], q[0]["d_bj"], q[1]["d_bj"], q[2]["d_bj"], g["d_vec"])
            for plane, p in enumerate(q):
                ref = self.d_cpic if plane else self.d_pic
                hv.pred_bi_d(p["taps"], bd, p["d_dst"], p["size"], ref, p["stride"], p["d_bj"].view(-1, 12), p["size"], p["size"])
                hv.satd_d(ref, p["stride"], p["d_dst"], p["size"], p["d_sj"], self.d_merge_satd[p["at"]:p["at"] + g["m"]], p["size"], p["size"])
            sat = [self.d_merge_satd[p["at"]:p["at"] + g["m"]] for p in q]
            hv.merge_decide_d(sat[0], sat[1], sat[2], len(g["sel"]), lam_q16, g["d_cost"], g["d_best"])
        self._merge = None

    @property
    def merge(self):
        """what merge_candidates() left on the device, as numpy: dict(vectors int16 [units, 5, 2 lists, 2], satd [units, 5, 3 planes], cost [units, 5] (Q16),
        best [units])"""
        if getattr(self, "_merge", None) is None:
            hv, u, K = self.hv, self.units, self.MERGE_CANDIDATES
            flat = hv.down(self.d_merge_satd, np.int32)
            vec, satd = np.zeros((len(u), K, 2, 2), np.int16), np.zeros((len(u), K, 3), np.int64)
            cost, best = np.zeros((len(u), K), np.int64), np.zeros(len(u), np.int32)
            for g in self.mgroups:
                for plane, p in enumerate(g["planes"]):
                    satd[g["sel"], :, plane] = flat[p["at"]:p["at"] + g["m"]].reshape(-1, K)
                vec[g["sel"]] = hv.down(g["d_vec"], np.int16).reshape(-1, K, 2, 2)
                with self.torch.cuda.stream(hv.tstream):
                    cost[g["sel"]] = g["d_cost"].cpu().numpy().reshape(-1, K)
                best[g["sel"]] = hv.down(g["d_best"], np.int32)
            self._merge = dict(vectors=vec, satd=satd, cost=cost, best=best)
        return self._merge

    def chroma_chain(self, field):
        """the inter residual of the chroma planes at the decided vectors (turing/Reconstruct.cpp:1274-1286 with cIdx 1, 2): HavocPredUni 4-tap of every unit's
        Cb and Cr block at its list-0 vector (eighth-sample phase = the luma vector's low three bits), then residual + DCT -> Rdoq::runQuantisation (cIdx) ->
        de-quantise + inverse DCT + add -> SSD, one chain per chroma transform size, into the chroma reconstruction planes.  Asynchronous."""
        hv, bd, torch, hmod = self.hv, self.bd, self.torch, self.hmod
        from . import workload
        u = self.units
        hw, hh = self.W // 2, self.H // 2
        if not hasattr(self, "cgroups"):
            self.cgroups = []
            for log2 in (5, 4, 3):
                sel = np.flatnonzero(u["log2_size"] == log2)
                if not len(sel):
                    continue
                cl, cn, m = log2 - 1, 1 << (log2 - 1), len(sel)
                x0, y0 = u["x0"][sel].astype(np.int64) // 2, u["y0"][sel].astype(np.int64) // 2
                qs, qshift, _ = workload.quant_params(self.qp, cl, bd, False)
                inv, dshift = workload.dequant_params(self.qp, cl, bd)
                lq, sf = hmod.rdoq_lambda(self.lam, inv)
                for comp in (1, 2):
                    fj = np.zeros((m, 4), np.int32)
                    fj[:, 0] = np.arange(m) * cn * cn
                    fj[:, 1] = 3 * (comp - 1) * self.cpe + (y0 + self.PAD // 2) * self.cstride + x0 + self.PAD // 2          # source block in d_cpic
                    fj[:, 2] = (comp - 1) * hw * hh + y0 * hw + x0                                                           # prediction block in cpred
                    fj[:, 3] = (comp - 1) * self.cpe + (y0 + self.PAD // 2) * self.cstride + x0 + self.PAD // 2              # reconstruction in crecon
                    jobs = np.zeros(m, hmod.RDOQ_JOB_DT)
                    jobs["dst_off"] = jobs["src_off"] = fj[:, 0]
                    jobs["quant_scale"], jobs["quant_shift"], jobs["inv_scale"], jobs["lambda_q16"], jobs["sdh_factor"] = qs, qshift, inv, lq, sf
                    jobs["sdh"], jobs["c_idx"] = 1, comp
                    jobs["ctx_index"] = (u["y0"][sel] // 64) * self.cx + u["x0"][sel] // 64
                    with torch.cuda.stream(hv.tstream):
                        d_rj = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(hv.device)
                    self.cgroups.append(dict(log2=cl, cn=cn, comp=comp, m=m, sel=sel, inv=inv, dshift=dshift, d_fj=hv.up(fj), d_rj=d_rj,
                                             d_x0=hv.up(u["x0"][sel].astype(np.int32)), d_y0=hv.up(u["y0"][sel].astype(np.int32)), d_dst=hv.up(fj[:, 2]),
                                             d_pj=hv.zeros(m * 8, np.int32), coef=hv.zeros(m * cn * cn, np.int16),
                                             level=hv.zeros(m * cn * cn, np.int16), cbf=hv.zeros(m, np.int32), ssd=hv.zeros(m, np.uint32), work=hv.rdoq_workspace(m)))
        for g in self.cgroups:
            hv.pred_jobs_d(self.layout, self.d_field, 0, g["d_x0"], g["d_y0"], g["log2"] + 1, g["comp"], g["d_dst"], g["d_pj"])
            hv.pred_uni_d(4, bd, self.cpred, hw, self.d_cpic, self.cstride, g["d_pj"].view(-1, 8), g["cn"], g["cn"])
            hv.tu_forward_d(bd, 0, g["log2"], g["coef"], self.d_cpic, self.cstride, self.cpred, hw, g["d_fj"])
            hv.rdoq_d(bd, g["log2"], g["level"], g["coef"], self.d_states, g["d_rj"], g["cbf"], g["work"])
            hv.tu_reconstruct_d(bd, 0, g["log2"], g["inv"], g["dshift"], self.crecon, self.cstride, self.cpred, hw, self.d_cpic, self.cstride, g["level"], g["d_fj"], g["ssd"])

    def tu_chain(self, field, predicted=False):
        """prediction at the decided vectors, then the residual-quadtree decisions and the reconstruction; returns (decisions, stats)"""
        if not predicted:
            self.predict(field)
        base = self.d_pic.data_ptr()
        self._rqt, st = rqt(self.hv.h, self.S, self.bd, base, self.origin, self.stride, self.pred.data_ptr(), self.W, self.recon.data_ptr(), self.origin,
                                   self.stride, self.d_states.data_ptr(), self.quant, self.lam, 1.0 / self.lam, self.units)
        self.rqt_stats = st
        return self._rqt, st

    # ---- round 4: the transform-tree decision and the block structure on the device -- nothing of the picture's step waits for the host after its searches ----
    def _rqt_plan(self):
        """what havoc_search_rqt builds per call, built ONCE (the units of a picture do not move): per transform size the candidates' job records (depth 0 of every unit
        of that size, then the four depth-1 blocks of every unit of the next larger size, unit by unit), their buffers, and where each unit finds its candidates"""
        hv, hmod, torch = self.hv, self.hmod, self.torch
        from . import workload
        u = self.units
        lists = {s: [] for s in (2, 3, 4, 5)}
        zero_at, one_at = np.zeros(len(u), np.int32), np.zeros(len(u), np.int32)
        for i in range(len(u)):
            L = int(u["log2_size"][i])
            zero_at[i] = len(lists[L])
            lists[L].append((i, 0, 0))
            one_at[i] = len(lists[L - 1])
            lists[L - 1] += [(i, 1, k) for k in range(4)]
        plan = dict(sizes={}, d_units=hv.up(np.ascontiguousarray(u).view(np.int32)), d_zero_at=hv.up(zero_at), d_one_at=hv.up(one_at),
                    d_out=hv.zeros(len(u) * 26, np.int32), table=np.zeros((4, 5), np.uint64), launches=0)
        for log2, cand in lists.items():
            m = len(cand)
            if not m:
                continue
            nn = 1 << log2
            area = nn * nn
            c = np.array(cand, np.int64)
            x = u["x0"][c[:, 0]].astype(np.int64) + np.where(c[:, 1] == 1, (c[:, 2] & 1) * nn, 0)
            y = u["y0"][c[:, 0]].astype(np.int64) + np.where(c[:, 1] == 1, (c[:, 2] >> 1) * nn, 0)
            jobs = np.stack([np.arange(m) * area, self.origin + y * self.stride + x, y * self.W + x, np.arange(m) * area], 1).astype(np.int32)
            qs, qshift, inv, dshift = (int(v) for v in self.quant[log2 - 2])
            rj = np.zeros(m, hmod.RDOQ_JOB_DT)
            rj["dst_off"] = rj["src_off"] = jobs[:, 0]
            rj["quant_scale"], rj["quant_shift"], rj["inv_scale"] = qs, qshift, inv
            rj["lambda_q16"], rj["sdh_factor"] = hmod.rdoq_lambda(self.lam, inv)
            rj["sdh"] = 1
            rj["ctx_index"] = u["ctx_index"][c[:, 0]]
            with torch.cuda.stream(hv.tstream):
                d_rj = torch.from_numpy(rj.view(np.uint8).reshape(-1)).to(hv.device)
            g = dict(log2=log2, nn=nn, m=m, inv=inv, dshift=dshift, d_jobs=hv.up(jobs), d_fin=hv.zeros(m * 4, np.int32), d_rj=d_rj,
                     d_sj=hv.up(np.stack([jobs[:, 0], np.full(m, area)], 1).astype(np.int32)), coef=hv.zeros(m * area, np.int16), level=hv.zeros(m * area, np.int16),
                     piece=hv.zeros(m * area, self.dt), work=hv.rdoq_workspace(m), cbf=hv.zeros(m, np.int32), ssd=hv.zeros(m, np.uint32), stats=hv.zeros(2 * m, np.int32),
                     ssd2=hv.zeros(m, np.uint32))
            plan["sizes"][log2] = g
            plan["table"][log2 - 2] = [g["cbf"].data_ptr(), g["ssd"].data_ptr(), g["stats"].data_ptr(), g["d_jobs"].data_ptr(), g["d_fin"].data_ptr()]
            plan["launches"] += 5
        plan["launches"] += 1
        # a block-sized area nobody reads, inside the reconstruction's bottom border (rewritten by the padding that ends the step): where the candidates that lost go
        plan["dump"] = (self.H + self.PAD + 16) * self.stride + self.PAD
        plan["rl_q16"] = int((1.0 / self.lam) * 65536 + 0.5)
        self._filter_buffers()
        return plan

    def _filter_buffers(self):
        """the block structure and the loop filter's edge data of the picture (one set per picture, whatever decides its units -- the whole-picture plan or the band views)"""
        if hasattr(self, "d_cells"):
            return
        hv, torch = self.hv, self.torch
        n = ((self.W + 63) // 64 * 8 + 1) * ((self.H + 63) // 64 * 8 + 1)
        with torch.cuda.stream(hv.tstream):
            self.d_data = torch.zeros(n, dtype=torch.int8, device=hv.device)
            self.d_bs = torch.zeros(n, dtype=torch.uint8, device=hv.device)
            self.d_cells = torch.zeros((self.H // 4) * (self.W // 4) * 16, dtype=torch.uint8, device=hv.device)
        self.d_chroma = hv.up(np.full(2 * (self.H // 2) * (self.W // 2), 128 << (self.bd - 8), self.dt))

    def tree_and_filter_on_device(self):
        """the transform-tree decisions of every inter unit (both depths through residual + DCT -> RDOQ -> IQ + IDCT + add -> SSD, the decision by k_rqt_decide, every
        candidate reconstructed again -- the chosen trees into the picture), the block structure (k_block_cells), boundary strengths, deblocking, padding: launches only"""
        hv, bd = self.hv, self.bd
        P = self.tree_decisions()
        hv.block_cells_d(self.W, self.H, self.qp, 0, self.d_field, P["d_units"].view(-1, 4), P["d_out"], self.d_cells)
        hv.derive_bs_d(self.d_cells, self.W // 4, self.W, self.H, self.d_data, self.d_bs)
        hv.deblock_d(bd, self.recon, self.origin, self.stride, self.d_chroma, 0, (self.H // 2) * (self.W // 2), self.W // 2, self.W, self.H, self.d_data, self.d_bs)
        hv.pad_block_d(self.recon, self.origin, self.W, self.H, self.stride, self.PAD)

    def tree_decisions(self):
        """the transform-tree part of tree_and_filter_on_device for self.units; returns the plan (its d_units / d_out are what the block structure is made from)"""
        hv, bd = self.hv, self.bd
        if not hasattr(self, "rqt_plan"):
            self.rqt_plan = self._rqt_plan()
        P = self.rqt_plan
        src = self.d_pic
        for g in P["sizes"].values():
            hv.tu_forward_d(bd, 0, g["log2"], g["coef"], src, self.stride, self.pred, self.W, g["d_jobs"].view(-1, 4))
            hv.rdoq_d(bd, g["log2"], g["level"], g["coef"], self.d_states, g["d_rj"], g["cbf"], g["work"])
            hv.tu_reconstruct_d(bd, 0, g["log2"], g["inv"], g["dshift"], g["piece"], g["nn"], self.pred, self.W, src, self.stride, g["level"], g["d_jobs"].view(-1, 4), g["ssd"])
            hv.level_stats_d(g["level"], g["d_sj"], g["m"], g["stats"])
        hv.rqt_decide_d(P["d_units"].view(-1, 4), P["d_zero_at"], P["d_one_at"], P["table"], self.origin, self.stride, P["dump"], P["rl_q16"], P["d_out"])
        for g in P["sizes"].values():
            hv.tu_reconstruct_d(bd, 0, g["log2"], g["inv"], g["dshift"], self.recon, self.stride, self.pred, self.W, src, self.stride, g["level"], g["d_fin"].view(-1, 4), g["ssd2"])
        return P

    @property
    def rqt_results(self):
        """the transform-tree decisions of the last step (RQT_RESULT_DT per unit): downloaded when asked for"""
        if getattr(self, "_rqt", None) is None:
            self._rqt = self.hv.down(self.rqt_plan["d_out"], np.int32).view(RQT_RESULT_DT).copy()
        return self._rqt

    @rqt_results.setter
    def rqt_results(self, v):
        self._rqt = v

    @property
    def cells(self):
        """the picture's block structure of the last step (CELL_DT [H / 4, W / 4]): on the device route k_block_cells leaves it in HBM and it is
        downloaded when asked for; the host route (search_on_device=False) made it on the host"""
        if getattr(self, "_cells", None) is None:
            from .havoc import CELL_DT
            self._cells = self.hv.down(self.d_cells, np.uint8).view(CELL_DT).reshape(self.H // 4, self.W // 4).copy()
        return self._cells

    @cells.setter
    def cells(self, v):
        self._cells = v

    def block_cells(self, field, decisions):
        """the picture's block structure after the decisions, as the 4x4 cells havoc_mi355x_derive_bs reads: every unit one inter 2Nx2N
        prediction unit from list 0 at the vector decided at its origin, its transform tree as decided (coded flags per block)"""
        from .havoc import CELL_DT
        cells = np.zeros((self.H // 4, self.W // 4), CELL_DT)
        field = np.ascontiguousarray(field)
        decisions = np.ascontiguousarray(decisions)
        rc = lib().havoc_search_block_cells(self.W, self.H, self.qp, 0, field.ctypes.data, self.units.ctypes.data, decisions.ctypes.data, len(self.units),
                                            cells.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"havoc_search_block_cells failed ({rc})")
        return cells

    def loop_filter(self, cells):
        """boundary strengths derived on the device from the block structure, in-loop deblocking of the reconstruction (luma; flat chroma planes
        stand in), padding: the reconstruction is then a reference picture.  Asynchronous."""
        hv, torch = self.hv, self.torch
        n = ((self.W + 63) // 64 * 8 + 1) * ((self.H + 63) // 64 * 8 + 1)
        if not hasattr(self, "d_bs"):
            with torch.cuda.stream(hv.tstream):
                self.d_data = torch.zeros(n, dtype=torch.int8, device=hv.device)
                self.d_bs = torch.zeros(n, dtype=torch.uint8, device=hv.device)
                self.d_cells = torch.zeros(cells.size * 16, dtype=torch.uint8, device=hv.device)
            self.d_chroma = hv.up(np.full(2 * (self.H // 2) * (self.W // 2), 128 << (self.bd - 8), self.dt))
        with torch.cuda.stream(hv.tstream):
            self.d_cells.copy_(torch.from_numpy(cells.view(np.uint8).reshape(-1)), non_blocking=True)
        hv.derive_bs_d(self.d_cells, cells.shape[1], self.W, self.H, self.d_data, self.d_bs)
        hv.deblock_d(self.bd, self.recon, self.origin, self.stride, self.d_chroma, 0, (self.H // 2) * (self.W // 2), self.W // 2, self.W, self.H, self.d_data, self.d_bs)
        hv.pad_block_d(self.recon, self.origin, self.W, self.H, self.stride, self.PAD)

    def tu_chain_fixed(self, field):
        """prediction at the decided list-0 vectors, then residual -> T -> RDOQ -> IQ -> IT + add -> SSD on fixed 16x16 (8x8) blocks; asynchronous"""
        hv, bd, pe = self.hv, self.bd, self.pe
        ref0 = self.d_pic[pe:2 * pe]
        src = self.d_pic[:pe]
        for g in self.groups:
            mv = field[0, g["y0"] >> 2, g["x0"] >> 2].astype(np.int64)          # quarter samples, [m, 2]
            pj = g["pj"]
            pj[:, 0] = g["y0"] * self.W + g["x0"]
            pj[:, 1] = (g["y0"] + (mv[:, 1] >> 2) + self.PAD) * self.stride + g["x0"] + (mv[:, 0] >> 2) + self.PAD
            pj[:, 2] = pj[:, 3] = g["nn"]
            pj[:, 4], pj[:, 5] = mv[:, 0] & 3, mv[:, 1] & 3
            with self.torch.cuda.stream(hv.tstream):
                g["d_pj"].copy_(self.torch.from_numpy(pj.reshape(-1)), non_blocking=True)
            hv.pred_uni_d(8, bd, self.pred, self.W, ref0, self.stride, g["d_pj"].view(-1, 8), g["nn"], g["nn"])
            hv.tu_forward_d(bd, 0, g["log2"], g["coef"], src, self.stride, self.pred, self.W, g["d_fj"])
            hv.rdoq_d(bd, g["log2"], g["level"], g["coef"], self.d_states, g["d_rj"], g["cbf"], g["work"])
            hv.tu_reconstruct_d(bd, 0, g["log2"], g["inv"], g["dshift"], self.recon, self.stride, self.pred, self.W, src, self.stride, g["level"], g["d_fj"], g["ssd"])

    def intra_decisions(self, on_device=True):
        """the intra side of the picture: per partition size the 35-mode SATD stage and its refinement order, then the RD refinement of every
        candidate mode and the champions.  on_device: the order, the candidates' job records and the champions are decided by kernels between the
        launches (havoc_search_intra_device: one call for all sizes, 40 bytes per partition come back); otherwise the two-call route with the
        decisions on the host (havoc_search_intra_modes + havoc_search_intra_rd; the tests' link to the per-call loops).
        Leaves {log2: (order or None, RD champions)} in self.intra_results and the calls' statistics in self.intra_stats"""
        hv = self.hv
        base = self.d_pic.data_ptr()
        out, self.intra_stats = {}, []
        if on_device:
            groups = [dict(log2=log2, n=len(g["jobs"]), d_nb=g["d_nb"].data_ptr(), d_jobs=g["d_jobs"].data_ptr(), d_ictx=g["d_ictx"].data_ptr(), d_ctu=g["d_ctu"].data_ptr(),
                           d_rec=g["d_rec"].data_ptr()) for log2, g in sorted(self.intra_parts.items(), reverse=True)]
            best, st = intra_device(hv.h, self.S, self.bd, base, self.stride, groups, self.d_states.data_ptr(), self.quant, self.rsl, self.lam, 1.0 / self.lam)
            out = {log2: (None, b) for log2, b in best.items()}
            self.intra_stats.append(st)
        else:
            for log2, g in sorted(self.intra_parts.items(), reverse=True):
                order = intra_modes(hv.h, self.S, self.bd, log2, base, self.stride, g["d_nb"].data_ptr(), g["d_jobs"].data_ptr(), len(g["jobs"]), g["ictx"], self.rsl)
                best, st = intra_rd(hv.h, self.S, self.bd, log2, base, self.stride, g["d_nb"].data_ptr(), g["jobs"], order, g["ictx"], g["ctu"], self.d_states.data_ptr(),
                                    self.quant[log2 - 2], self.lam, 1.0 / self.lam, g["d_rec"].data_ptr())
                st.launches += 1      # the 35-mode stage
                out[log2] = (order, best)
                self.intra_stats.append(st)
        self.intra_results = out
        return out

    def _replayed(self, key, fn):
        """run a FIXED sequence of launches (same kernels, same device buffers every picture: the job tables that depend on the decisions are made on the
        device) -- the first time as it is (it allocates), the second time recorded into a HIP graph, from then on as one graph launch: a picture's ~70
        launches after its searches cost their issuing thread ~100 us each with 8 pictures in flight (profiles/r04/inflight8_timeline.txt)"""
        state = self._graphs.get(key)
        if not self.use_graphs or state is None:
            fn()
            self._graphs[key] = False
        else:
            if state is False:
                state = self._graphs[key] = self.hv.graph_capture(fn)
            self.hv.graph_launch(state)

    def step(self):
        self.phase_planes()
        res, field, stats = self.search()
        if self.intra_parts:
            # (running this on a second context / stream beside the searches -- it reads nothing they decide -- was measured: one picture alone 15.7 -> 15.5 ms, but
            # 294 -> 268 / 383 -> 237 pictures/s with 8 / 16 in flight: a second issuing thread per picture costs more than the overlap gives)
            self.intra_decisions()
        self._merge = None
        if self.search_on_device:
            # everything after the searches is a FIXED sequence of launches over device-resident tables (the decided field never leaves the device, the decisions
            # between the launches are kernels): recorded once into a HIP graph, one launch per picture, one wait at the end
            self._rqt = self._cells = None
            self._replayed("after the searches", lambda: (self.merge_candidates(field), self.predict(field), self.tree_and_filter_on_device(), self.chroma_chain(field)))
            self.rqt_stats = RqtStats()
            self.rqt_stats.launches, self.rqt_stats.candidates = self.rqt_plan["launches"], 5 * len(self.units)
        else:
            self._replayed("merge + predict", lambda: (self.merge_candidates(field), self.predict(field)))
            decisions, _ = self.tu_chain(field, predicted=True)
            self._replayed("chroma", lambda: self.chroma_chain(field))
            self.cells = self.block_cells(field, decisions)
            self.loop_filter(self.cells)
        self.hv.sync()
        return res, field, stats

    # ---- round 5: the PRODUCER's half of CTU-row bands (turing/TaskDeblock.cpp:151-167: a picture's rows are deblocked, padded and published while the rows below are
    # still being encoded; the consumer's half is havoc_mi355x_search_gate) ---------------------------------------------------------------------------------------
    def _band_views(self, band_ctu_rows, side):
        """the picture's units cut into bands of whole CTU rows: per band a shallow copy of this object that owns the band's units, job tables and buffers and shares the
        picture's planes, field and block structure; its launches go to `side` (a Havoc context on a stream of another priority than the searches')"""
        import copy
        key = (band_ctu_rows, id(side))
        if getattr(self, "_views_key", None) != key:
            self._filter_buffers()
            rows = band_ctu_rows * 64
            views = []
            for y0 in range(0, self.H, rows):
                v = copy.copy(self)
                for name in ("mgroups", "pgroups", "cgroups", "rqt_plan", "d_merge_satd", "_merge", "_rqt", "_cells", "_views", "_views_key", "_graphs"):
                    v.__dict__.pop(name, None)
                v.hv, v._graphs, v.use_graphs = side, {}, True
                v.units = self.units[(self.units["y0"] >= y0) & (self.units["y0"] < y0 + rows)]
                v.band_span = (y0, min(self.H, y0 + rows))
                views.append(v)
            self._views, self._views_key = views, key
        return self._views

    def band_rows(self, band_ctu_rows):
        """[first, last) luma rows of the bands step_banded cuts the picture into"""
        rows = band_ctu_rows * 64
        return [(y0, min(self.H, y0 + rows)) for y0 in range(0, self.H, rows)]

    def step_banded(self, side, band_ctu_rows=4, rows_final=None, on_band=None, before_band=None, on_queued=None, make_phase_planes=True, wait=True):
        """step() with everything after the searches done BAND BY BAND while the rows below are still searched: the search kernel is launched on this picture's stream
        and nothing waits for it on the host; on `side` (a Havoc context on a stream of ANOTHER PRIORITY, so that it has its own hardware queue) every band of
        band_ctu_rows CTU rows is queued behind a launch that ends when the band's rows and the row below them are searched (havoc_mi355x_search_wait_rows; the merge
        candidate below-left of a unit lies in the next CTU row): merge candidates, prediction, transform trees, chroma, then the band's block structure, boundary
        strengths, deblocking (its own edges and the edge on its top: the three rows above it change once more) and padding.  rows_final (optional int32 device tensor
        of 1): raised after every band to the number of luma rows of the reconstruction that are final, padding included -- what a dependent picture's search_gate can be
        built on (after the filtering of its taps' rows).  on_band(b, final_rows) (optional) is called when band b's launches are queued: what it queues on `side`
        runs when the band is final (hand the rows on to a dependent picture: tests/test_step_banded.py); before_band(b) is called before band b's launches are queued
        (what it queues on `side` runs before them: bring in the rows of the references the band reads); on_queued() when everything of the picture is queued and
        nothing has been waited for.  make_phase_planes = False: the fractional planes of the references are somebody else's business (a pipeline whose follow steps make
        them band by band: a whole-plane pass here would race with those).  wait = False: nothing is waited for and nothing downloaded -- the call only queues (the intra
        candidates, whose call is synchronous, are then left out); the caller orders what follows with events.  Same results as step(), returned the same way."""
        hv, torch, W, H = self.hv, self.torch, self.W, self.H
        if not self.search_on_device:
            raise ValueError("step_banded needs the device search")
        views = self._band_views(band_ctu_rows, side)
        if make_phase_planes:
            self.phase_planes()
        if not hasattr(self, "_dsearch"):
            n = len(self.pus)
            self._dsearch = dict(pus=hv.up(np.ascontiguousarray(self.pus).view(np.uint8).reshape(-1)), first=hv.up(np.ascontiguousarray(self.ctu_first, np.int32)),
                                 out=hv.zeros(2 * n * RESULT_DT.itemsize, np.uint8), bi=hv.zeros(2 * n * RESULT_DT.itemsize, np.uint8),
                                 work=hv.zeros((self.hmod.search_workspace(W, H) + 3) // 4, np.int32), gave_up=hv.zeros(1, np.int32))
        D = self._dsearch
        with torch.cuda.stream(hv.tstream):
            D["work"].zero_()      # (the rows' progress counters of the previous picture: gone before anything on `side` can read them)
            D["gave_up"].zero_()
            if rows_final is not None:
                rows_final.zero_()
            cleared = torch.cuda.Event()
            cleared.record(hv.tstream)
        pe, o = self.pe, self.origin
        hv.search_picture_uni_d(self.S, self.params, self.mvp_rate, self.d_pic, o, self.stride, self.d_pic, (pe + o, 2 * pe + o), self.stride, self.d_phase, pe, (o, 16 * pe + o),
                                D["pus"], D["first"], self.cx, self.cy, len(self.pus), D["out"], D["bi"], self.d_field, D["work"])
        side.tstream.wait_event(cleared)
        none = side.zeros(4, np.int32).view(-1, 4)[:0]
        side.block_cells_d(W, H, self.qp, 0, self.d_field, none, none, self.d_cells)      # blank cells, once; the bands add theirs
        bstride = (W + 63) // 64 * 8 + 1
        half = (H // 2) * (W // 2)
        for b, v in enumerate(views):
            y0, y1 = v.band_span
            last = b == len(views) - 1
            if before_band is not None:
                before_band(b)

            def band(v=v, b=b, y0=y0, y1=y1, last=last):
                side.search_wait_rows(D["work"], W, H, min(self.cy - 1, (y1 - 1) // 64 + 1), D["gave_up"])
                v.merge_candidates(None)
                v.predict(None)
                P = v.tree_decisions()
                v.chroma_chain(None)
                side.block_cells_add_d(W, H, self.qp, 0, self.d_field, P["d_units"].view(-1, 4), P["d_out"], self.d_cells)
                side.derive_bs_d(self.d_cells, W // 4, W, H, self.d_data, self.d_bs)
                side.deblock_d(self.bd, self.recon, self.origin + y0 * self.stride, self.stride, self.d_chroma, (y0 // 2) * (W // 2), half + (y0 // 2) * (W // 2), W // 2, W, y1 - y0,
                               self.d_data[(y0 // 8) * bstride:], self.d_bs[(y0 // 8) * bstride:])
                lo = max(0, y0 - 8)
                side.pad_block_d(self.recon, self.origin + lo * self.stride, W, y1 - lo, self.stride, self.PAD, top=b == 0, bottom=last)

            # a band's ~170 launches are the same every picture (its job tables are made on the device): one graph launch from the third picture on -- with eight pictures
            # in flight a launch costs its issuing thread ~100 us (the plain form ran one 1080p sequence at 45 pictures/s instead of 108)
            v._replayed("band", band)
            final = H + self.PAD if last else y1 - 4
            if rows_final is not None:
                with torch.cuda.stream(side.tstream):
                    rows_final.fill_(final)
            if on_band is not None:
                on_band(b, final)
        if on_queued is not None:
            on_queued()
        if not wait:
            return None      # the caller MUST call check_banded() after its own synchronisation: a wait that gave up leaves a wrong reconstruction and no error
        if self.intra_parts:
            self.intra_decisions()
        hv.sync()
        side.sync()
        self.check_banded()
        self._merge = self._rqt = self._cells = None
        res = hv.down(D["out"], np.uint8).view(RESULT_DT)
        self.bi_results = hv.down(D["bi"], np.uint8).view(RESULT_DT)
        field = hv.down(self.d_field, np.int16).reshape(2, (H + 3) // 4, (W + 3) // 4, 2)
        return res, field, None

    def check_banded(self):
        """after step_banded (wait=False: after the caller's own synchronisation of both streams): raises if a device-side wait on the search's progress gave up (a side
        stream sharing a hardware queue with the search it waits for, ~4-8 s) -- the band's launches then ran on unfinished rows"""
        D = getattr(self, "_dsearch", None)
        if D is None:
            return
        if int(self.hv.down(D["gave_up"], np.int32)[0]) or int(self.hv.down(D["work"], np.int32)[-1]):
            raise RuntimeError("step_banded: a wait on the search's progress gave up; the picture's reconstruction is not to be used")

    def results(self):
        """what the TU chain left on the device, as numpy (per group): coefficients, levels, flags, SSDs; and the reconstruction"""
        hv = self.hv
        out = [dict(log2=g["log2"], coef=hv.down(g["coef"], np.int16), level=hv.down(g["level"], np.int16), cbf=hv.down(g["cbf"], np.int32),
                    ssd=hv.down(g["ssd"], np.uint32)) for g in self.groups]
        return out, hv.down(self.recon, self.dt)
