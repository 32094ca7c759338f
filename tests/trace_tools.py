"""TEST INFRASTRUCTURE: the reference encoder's own decision loops as the checker of turingcodec_amd/search/decision.hpp.

oracle/_ref/turing_ref_trace is the reference encoder (the same objects as turing_ref_havoc) whose Search.hpp carries the trace points of
oracle/trace_hooks.h: for every searchMotionUni / searchMotionBi / searchIntraPartition of an encode it writes the inputs as the reference's code
holds them, every primitive call the loop makes (positions and returned values) and what it decided.  This module

  * runs that encoder on a seeded clip (stream checked against the committed hash by the caller: the trace points change nothing),
  * parses the trace into the records of turingcodec_amd/search/search_abi.h (havoc_search_pu inputs, havoc_search_result decisions) plus the
    call sequence of every search,
  * rebuilds the pictures the searches ran on: the source frames of the clip and the encoder's own reconstructed pictures (--dump-pictures),
    padded by replication as turing/Padding.h pads them,

so that tests/test_trace_pin.py can run decision.hpp -- on the CPU through the reference's tables, and on the MI355X inside the search kernel --
on the reference's inputs and require the reference's call sequence and decisions.
"""
import os

import numpy as np

import encoder_tools as et
import search_tools as st

TRACE_EXE = os.path.join(et.REFDIR, "turing_ref_trace")
REC_DT = np.dtype([("thread", "u4"), ("kind", "u2"), ("n", "u2"), ("v", "i4", (14,))])
assert REC_DT.itemsize == 64
(UNI_BEGIN, BEGIN2, SAD, SAD4, SATD, UNI_INTEGER, UNI_SUBPEL, UNI_END, BI_BEGIN, BI_MV, BI_END, INTRA_BEGIN, INTRA_SATD, INTRA_MAX, INTRA_PICK, INTRA_SSD,
 INTRA_END, INTRA_RATE, INTRA_SWAP, RQT_ONE, RQT_ZERO, RQT_END) = range(1, 23)
AMVP, AMVP_NB = 25, 26         # (round 5) predictMvp's inputs and outputs per searchUni call
COL, COL_PU = 32, 33          # the collocated picture's two cells a temporal candidate can come from, after an AMVP / MERGE group
MERGE, MERGE_NB, MERGE_COL, MERGE_POC, MERGE_OUT = 27, 28, 29, 30, 31      # populateMergeCandidates' inputs and the list it left, per searchMergeModes call
INTRA_NB, INTRA_NBF = 23, 24      # (round 5) the partition's reference samples as the encoder held them: unfiltered / filtered, 14 per record, before INTRA_BEGIN
PAD = 96


def have_trace_encoder():
    return os.path.exists(TRACE_EXE)


def i64(lo, hi):
    return (lo.astype(np.int64) & 0xFFFFFFFF) | (hi.astype(np.int64) << 32)


def f64(lo, hi):
    return i64(lo, hi).view(np.float64)


def run(case, workdir, threads=1):
    """encode `case` (encoder_tools.CASES) with the traced encoder: (stream bytes, trace records in call order, reconstructed pictures [frames][3 planes])"""
    w, h, n, seed, bd, opts = et.CASES[case]
    trace = os.path.join(workdir, case + ".trace")
    rec = os.path.join(workdir, case + ".rec.yuv")
    stream, _ = et.encode(TRACE_EXE, case, workdir, ["--threads", str(threads), "--dump-pictures", rec], env={"HAVOC_TRACE_FILE": trace}, tag=".trace")
    records = np.fromfile(trace, REC_DT)
    internal = bd
    if "--internal-bit-depth" in opts:
        internal = int(opts[opts.index("--internal-bit-depth") + 1])
    dt = np.uint8 if internal == 8 else np.uint16
    raw = np.fromfile(rec, dt)
    per = w * h * 3 // 2
    assert raw.size == per * n, (raw.size, per, n)
    frames = []
    for k in range(n):
        f = raw[k * per:(k + 1) * per]
        frames.append((f[:w * h].reshape(h, w), f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), f[w * h * 5 // 4:].reshape(h // 2, w // 2)))
    os.remove(trace)
    return stream, records, frames, internal


def source_frames(case, internal):
    """the clip's frames as the encoder holds them (8-bit input widened by << (internal - 8) when the encoder runs 16-bit samples, turing/encode.cpp:341-449)"""
    from turingcodec_amd import workload
    w, h, n, seed, bd, _ = et.CASES[case]
    out = []
    for planes in workload.synth_frames(w, h, n, seed, bit_depth=bd):
        if internal > bd:
            planes = [p.astype(np.uint16) << (internal - bd) for p in planes]
        out.append(planes)
    return out


def padded(plane, pad=PAD):
    """a picture plane with `pad` replicated samples around it (turing/Padding.h), flat, 64-byte aligned; (array, stride)"""
    p = np.pad(plane, pad, mode="edge")
    raw = np.empty(p.size * p.itemsize + 64, np.uint8)
    o = (-raw.ctypes.data) % 64
    out = raw[o:o + p.size * p.itemsize].view(p.dtype)
    out[...] = p.ravel()
    return out, p.shape[1]


def _segments(kind, begin, end):
    """indices of the begin / end records of the non-nested segments of one thread's record stream"""
    b, e = np.flatnonzero(kind == begin), np.flatnonzero(kind == end)
    assert len(b) == len(e) and np.all(b < e) and np.all(b[1:] > e[:-1]), "trace segments are not properly paired"
    return b, e


def _call_rows(rec):
    """the sad / sad4 / satd records as rows of the call log tests/search_client.cpp writes: kind, x0, y0 .. x3, y3, value0..3"""
    rows = np.zeros((len(rec), 13), np.int32)
    k = rec["kind"]
    rows[:, 0] = k
    one = (k == SAD) | (k == SATD)
    rows[one, 1:3] = rec["v"][one, 0:2]
    rows[one, 9] = rec["v"][one, 2]
    four = k == SAD4
    rows[four, 1:13] = rec["v"][four, 0:12]
    return rows


class MotionTrace:
    """every searchMotionUni (bi=False) or searchMotionBi (bi=True) of the encode, in the order the encoder ran them"""

    def __init__(self, records, bi=False):
        begin, end = (BI_BEGIN, BI_END) if bi else (UNI_BEGIN, UNI_END)
        pus, meta, results, rows, first = [], [], [], [], [0]
        for t in np.unique(records["thread"]):
            mine = np.flatnonzero(records["thread"] == t)
            rec = records[mine]
            kind = rec["kind"].astype(np.int32)
            b, e = _segments(kind, begin, end)
            if not len(b):
                continue
            v = rec["v"]
            a, a2 = v[b], v[b + 1]
            assert np.all(kind[b + 1] == BEGIN2)
            pu = np.zeros(len(b), st.PU_DT)
            pu["x0"], pu["y0"], pu["w"], pu["h"] = a[:, 3], a[:, 4], a[:, 5], a[:, 6]
            pu["cu_log2_size"], pu["cqt_depth"], pu["part_2Nx2N"], pu["ref_list"] = a[:, 7], a[:, 8], a[:, 9], a[:, 2]
            pu["x_ctb"], pu["y_ctb"] = a[:, 10], a[:, 11]
            pu["mvp"] = a2[:, 0:4].reshape(-1, 2, 2)
            pu["mv_previous_2Nx2N"] = a2[:, 4:6]
            pu["mvp_rate"][:, 0], pu["mvp_rate"][:, 1] = i64(a2[:, 6], a2[:, 7]), i64(a2[:, 8], a2[:, 9])
            m = np.zeros(len(b), [("poc", "i4"), ("ref_poc", "i4"), ("concurrent_frames", "i4"), ("flags", "i4"), ("rsl", "f8"), ("bit_depth", "i4"),
                                  ("ctb", "i4"), ("thread", "i4"), ("seq", "i8"), ("start", "i2", (2,))])
            m["seq"] = mine[b]                                             # position in the trace file = the order the encoder ran the searches in
            m["poc"], m["ref_poc"], m["concurrent_frames"], m["flags"] = a[:, 0], a[:, 1], a[:, 12], a[:, 13]
            m["rsl"], m["bit_depth"], m["ctb"], m["thread"] = f64(a2[:, 10], a2[:, 11]), a2[:, 12], a2[:, 13], t
            res = np.zeros(len(b), st.RESULT_DT)
            r_end = v[e]
            if bi:
                assert np.all(kind[b + 2] == BI_MV)
                mv = v[b + 2][:, 0:4].reshape(-1, 2, 2)                     # [search][list][x, y]
                lst = pu["ref_list"]
                idx = np.arange(len(b))
                m["start"] = mv[idx, lst]
                pu["mv_other"] = mv[idx, 1 - lst]
                res["mv"], res["mvd"], res["mvp_flag"] = r_end[:, 0:2], r_end[:, 2:4], r_end[:, 4]
                res["cost_subpel"] = i64(r_end[:, 5], r_end[:, 6])
            else:
                # UNI_INTEGER is the record after the last integer call; UNI_SUBPEL (if the speed refines) the one before UNI_END
                is_int = np.flatnonzero(kind == UNI_INTEGER)
                assert len(is_int) == len(b) and np.all((is_int > b) & (is_int < e))
                ri = v[is_int]
                res["mv_integer"], res["mvp_flag"], res["cost_integer"] = ri[:, 0:2], ri[:, 4], i64(ri[:, 5], ri[:, 6])
                res["mv"], res["mvd"] = ri[:, 0:2], ri[:, 2:4]
                sub = kind[e - 1] == UNI_SUBPEL
                res["mv"][sub], res["mvd"][sub] = v[e - 1][sub, 0:2], v[e - 1][sub, 2:4]
                assert np.array_equal(res["mvd"], r_end[:, 0:2]) and np.array_equal(res["mvp_flag"], r_end[:, 2])
            call = (kind == SAD) | (kind == SAD4) | (kind == SATD)
            inside = (np.cumsum(kind == begin) - np.cumsum(kind == end)) > 0
            csum = np.concatenate([[0], np.cumsum(call & inside)])
            counts = csum[e] - csum[b]
            res["calls"] = counts
            rows.append(_call_rows(rec[call & inside]))
            first.extend((first[-1] + np.cumsum(counts)).tolist())
            pus.append(pu)
            meta.append(m)
            results.append(res)
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
        self.pus, self.meta, self.results = cat(pus, st.PU_DT), cat(meta, np.dtype([("poc", "i4")])), cat(results, st.RESULT_DT)
        self.rows = np.concatenate(rows) if rows else np.zeros((0, 13), np.int32)
        self.first = np.asarray(first, np.int64)
        self.bi = bi

    def __len__(self):
        return len(self.pus)

    def groups(self):
        """searches that share pictures and encoder settings: {(poc, ref_poc, other_poc or -1 ...): indices}; one client call each"""
        m = self.meta
        keys = np.stack([m["poc"], m["ref_poc"], m["concurrent_frames"], m["flags"], m["bit_depth"], m["ctb"], m["rsl"].view(np.int64) & 0xFFFFFFFF,
                         m["rsl"].view(np.int64) >> 32], axis=1)
        out = {}
        uniq, inv = np.unique(keys, axis=0, return_inverse=True)
        for g in range(len(uniq)):
            out[tuple(int(x) for x in uniq[g])] = np.flatnonzero(inv.ravel() == g)
        return out

    def params(self, key, width, height):
        poc, ref_poc, cf, flags, bd, ctb = key[:6]
        rsl = float(np.array([key[6] | (key[7] << 32)], np.int64).view(np.float64)[0])
        return st.Params(width, height, ctb, cf, flags & 1, (flags >> 1) & 1, (flags >> 2) & 1, (flags >> 3) & 1, (flags >> 4) & 1, bd, rsl)


class IntraTrace:
    """every searchIntraPartition: the 35 distortions predictIntraLuma returned, the rate offsets, and the order the modes went to RD refinement in"""

    def __init__(self, records):
        ctx, satd, costs, order, count, rsl, where, champion = [], [], [], [], [], [], [], []
        mode_ab = []
        nb, nbf = [], []            # per partition: the unfiltered / filtered reference samples (None when the encoder had none: 64x64 / 4x4 filtered)
        cand, rl = [], []          # RD refinement: per candidate (mode, ssd, rate or -1 when the encoder did not measure it); reciprocalLambda (Q16) per partition
        for t in np.unique(records["thread"]):
            rec = records[records["thread"] == t]
            kind = rec["kind"].astype(np.int32)
            b, e = _segments(kind, INTRA_BEGIN, INTRA_END)
            v = rec["v"]
            for k in range(len(b)):
                seg_kind, seg = kind[b[k]:e[k] + 1], v[b[k]:e[k] + 1]
                a = seg[0]
                # the reference samples of the partition: the NB / NBF records that directly precede the BEGIN record
                at = b[k]
                while at > 0 and kind[at - 1] in (INTRA_NB, INTRA_NBF):
                    at -= 1
                parts = {INTRA_NB: [], INTRA_NBF: []}
                for q in range(at, b[k]):
                    parts[int(kind[q])].append(v[q][:int(rec["n"][q])])
                length = 4 * (1 << int(a[3])) + 1
                for key_, store in ((INTRA_NB, nb), (INTRA_NBF, nbf)):
                    arr = np.concatenate(parts[key_]).astype(np.int32) if parts[key_] else None
                    assert arr is None or len(arr) == length, (len(arr), length)
                    store.append(arr)
                s = seg[seg_kind == INTRA_SATD]
                assert len(s) == 35 and np.array_equal(s[:, 0], np.arange(35))
                c = np.zeros(1, st.INTRA_CTX_DT)[0]
                c["cand_mode_list"], c["neighbour_modes"] = a[4:7], int(a[7]) & 0xFF
                mode_ab.append([(int(a[7]) >> 8) & 0xFF, (int(a[7]) >> 16) & 0xFF])      # candIntraPredModeA (left), B (above) as CandModeList::getCandidate returned them
                c["max_refine"] = seg[seg_kind == INTRA_MAX][0][0]
                c["rate_a_minus_c"], c["rate_b_minus_c"] = i64(a[8:9], a[9:10])[0], i64(a[10:11], a[11:12])[0]
                picks = seg[seg_kind == INTRA_PICK]
                assert np.array_equal(picks[:, 0], np.arange(len(picks)))
                o = np.full(35, -1, np.int32)
                o[:len(picks)] = picks[:, 1]
                ctx.append(c)
                satd.append(s[:, 1])
                costs.append(i64(s[:, 2], s[:, 3]))
                order.append(o)
                count.append(len(picks))
                rsl.append(f64(a[12:13], a[13:14])[0])
                where.append(a[0:4])
                champion.append(seg[-1][0])
                # the refinement loop's records per candidate: PICK, [nested searches of reconstructIntraLuma: none], SSD, [RATE], [SWAP]
                pk = np.flatnonzero(seg_kind == INTRA_PICK)
                rows = np.full((len(pk), 3), -1, np.int64)
                lam = 0
                for n_, at in enumerate(pk):
                    end = pk[n_ + 1] if n_ + 1 < len(pk) else len(seg_kind) - 1
                    kinds = seg_kind[at:end]
                    rows[n_, 0] = seg[at][1]
                    ss = np.flatnonzero(kinds == INTRA_SSD)
                    assert len(ss) == 1
                    rows[n_, 1], lam = seg[at + ss[0]][0], seg[at + ss[0]][1]
                    rr = np.flatnonzero(kinds == INTRA_RATE)
                    if len(rr):
                        rows[n_, 2] = i64(seg[at + rr[0]][0:1], seg[at + rr[0]][1:2])[0]
                cand.append(rows)
                rl.append(lam)
        self.ctx = np.array(ctx, st.INTRA_CTX_DT) if ctx else np.zeros(0, st.INTRA_CTX_DT)
        self.mode_ab = np.array(mode_ab, np.int32).reshape(-1, 2)      # the neighbours' modes the candModeList was made from
        self.satd = np.array(satd, np.int32).reshape(-1, 35)
        self.costs = np.array(costs, np.int64).reshape(-1, 35)
        self.order = np.array(order, np.int32).reshape(-1, 35)
        self.count = np.array(count, np.int32)
        self.rsl = np.array(rsl, np.float64)
        self.where = np.array(where, np.int32).reshape(-1, 4)      # poc, x, y, log2 partition size
        self.champion = np.array(champion, np.int32)
        self.candidates = np.concatenate(cand) if cand else np.zeros((0, 3), np.int64)
        self.reciprocal_lambda = np.array(rl, np.int32)
        self.neighbours, self.neighbours_filtered = nb, nbf

    def __len__(self):
        return len(self.ctx)


class AmvpTrace:
    """every predictMvp of searchUni (turing/Search.hpp:1779 -> Mvp.h:195-436): the five spatial neighbours as neighbourPuData() returned them, the temporal candidate,
    the two predictors the encoder derived.  rows = the inputs in the layout of tests/search_client.cpp: client_amvp; mvp = int32 [n, 4]"""

    def __init__(self, records):
        rows, mvp, where, temporal, temporal_want, geometry, position = [], [], [], [], [], [], []
        for t in np.unique(records["thread"]):
            rec = records[records["thread"] == t]
            kind = rec["kind"].astype(np.int32)
            v = rec["v"]
            for i in np.flatnonzero(kind == AMVP):
                assert np.all(kind[i + 1:i + 6] == AMVP_NB) and np.array_equal(v[i + 1:i + 6, 0], np.arange(5)), "an AMVP record is followed by its five neighbours"
                a = v[i]
                r = np.zeros(52, np.int32)
                r[0], r[1], r[2] = a[1], a[0], a[7]
                for k in range(5):
                    r[4 + 9 * k:13 + 9 * k] = v[i + 1 + k][1:10]
                r[49], r[50], r[51] = a[8], a[9], a[10]
                rows.append(r)
                s16 = lambda u: u - 65536 if u >= 32768 else u
                unpack = lambda p: (s16(p & 0xFFFF), s16((p >> 16) & 0xFFFF))
                mvp.append(unpack(int(a[11]) & 0xFFFFFFFF) + unpack(int(a[12]) & 0xFFFFFFFF))
                where.append(a[0:7])
                dims = int(a[13]) & 0xFFFFFFFF
                geometry.append([a[3], a[4], a[5], a[6], 64, dims & 0xFFFF, dims >> 16, 0])
                position.append(v[i + 1:i + 6, 10])
                if i + 8 < len(kind) and kind[i + 6] == COL:      # the collocated picture's cells: the temporal candidate of (list a[1], target POC a[7]) must come out of them
                    c = v[i + 6]
                    t = np.zeros(36, np.int32)
                    t[0:6] = [a[1], c[10], a[7], c[4], c[5], c[6]]
                    t[6:13] = [c[0], c[1], c[2], c[3], c[7], c[8], c[9]]
                    t[14:24], t[24:34] = v[i + 7][1:11], v[i + 8][1:11]
                    temporal.append(t)
                    temporal_want.append([a[8], a[9] if a[8] else 0, a[10] if a[8] else 0])
        self.rows = np.array(rows, np.int32).reshape(-1, 52)
        self.mvp = np.array(mvp, np.int32).reshape(-1, 4)
        self.where = np.array(where, np.int32).reshape(-1, 7)
        self.geometry = np.array(geometry, np.int32).reshape(-1, 8)             # inputs of client_positions_available
        self.position = np.array(position, np.int32).reshape(-1, 5)             # A0, A1, B0, B1, B2: the encoder's neighbourPuData would look at what is stored there
        self.temporal = np.array(temporal, np.int32).reshape(-1, 36)            # inputs of client_temporal
        self.temporal_want = np.array(temporal_want, np.int32).reshape(-1, 3)   # available, x, y as the encoder derived them

    def __len__(self):
        return len(self.rows)


class MergeTrace:
    """every populateMergeCandidates of searchMergeModes (turing/Search.hpp:1763 -> Mvp.h:486-697): the five spatial neighbours as PuMergeNeighbour<>::get returned them,
    the temporal candidate, the reference lists' picture order counts, and the list the encoder left.  rows = the inputs in the layout of tests/search_client.cpp:
    client_merge (int32 [n, 64]); out = int32 [n, 5, 8]; where = poc, xPb, yPb, nPbW, nPbH"""

    def __init__(self, records):
        rows, outs, where, temporal, temporal_want = [], [], [], [], []
        for t in np.unique(records["thread"]):
            rec = records[records["thread"] == t]
            kind = rec["kind"].astype(np.int32)
            v = rec["v"]
            for i in np.flatnonzero(kind == MERGE):
                a = v[i]
                ncand = min(int(a[9]), 5)
                assert np.all(kind[i + 1:i + 6] == MERGE_NB) and np.array_equal(v[i + 1:i + 6, 0], np.arange(5)) and kind[i + 6] == MERGE_COL and kind[i + 7] == MERGE_POC
                assert np.all(kind[i + 8:i + 8 + ncand] == MERGE_OUT) and np.array_equal(v[i + 8:i + 8 + ncand, 0], np.arange(ncand)), "a MERGE record is followed by its list"
                r = np.zeros(64, np.int32)
                r[0:8] = [a[5], a[3], a[4], a[6], a[7], a[8], a[9], a[11]]
                for k in range(5):
                    r[8 + 8 * k:16 + 8 * k] = v[i + 1 + k][2:10]
                r[48:56] = v[i + 6][2:10]
                r[56:64] = v[i + 7][0:8]
                o = np.zeros((5, 8), np.int32)
                for k in range(ncand):
                    o[k] = v[i + 8 + k][2:10]
                rows.append(r)
                outs.append(o)
                where.append(a[0:5])
                j = i + 8 + ncand
                if a[10] and j + 2 < len(kind) and kind[j] == COL:      # temporal candidates enabled: for list 0 and, in a B slice, list 1, towards reference index 0
                    c = v[j]
                    for X in range(2 if a[6] else 1):
                        t = np.zeros(36, np.int32)
                        t[0:6] = [X, c[10], r[56 + 4 * X], c[4], c[5], c[6]]
                        t[6:13] = [c[0], c[1], c[2], c[3], c[7], c[8], c[9]]
                        t[14:24], t[24:34] = v[j + 1][1:11], v[j + 2][1:11]
                        temporal.append(t)
                        have = int(r[48 + X])
                        temporal_want.append([have, r[52 + 2 * X] if have else 0, r[53 + 2 * X] if have else 0])
        self.temporal = np.array(temporal, np.int32).reshape(-1, 36)
        self.temporal_want = np.array(temporal_want, np.int32).reshape(-1, 3)
        self.rows = np.array(rows, np.int32).reshape(-1, 64)
        self.out = np.array(outs, np.int32).reshape(-1, 5, 8)
        self.where = np.array(where, np.int32).reshape(-1, 5)

    def __len__(self):
        return len(self.rows)


class RqtTrace:
    """every residual-quadtree decision of reconstructInter (turing/Reconstruct.cpp:1296-1428) that had a choice: the two candidates' distortions (three planes)
    and RATES as the encoder's entropy estimator gave them, and the depth it chose"""

    def __init__(self, records):
        rows, chosen, uncoded = [], [], 0
        for t in np.unique(records["thread"]):
            rec = records[records["thread"] == t]
            kind = rec["kind"].astype(np.int32)
            v = rec["v"]
            end = np.flatnonzero(kind == RQT_END)
            one = np.flatnonzero(kind == RQT_ONE)
            # a decision = [RQT_ONE .. RQT_ZERO ..] RQT_END; the ONE / ZERO records of a decision directly precede its END (nothing of another
            # decision in between: reconstructInter does not recurse into itself); units whose split tree was uncoded have only the END
            last_one = {}
            for i in one:
                j = end[np.searchsorted(end, i)]
                last_one[j] = i
            for j in end:
                depth, cbf_zero = int(v[j][0]), int(v[j][1])
                if cbf_zero:
                    assert j not in last_one
                    uncoded += 1
                    continue
                i = last_one[j]
                z = i + np.flatnonzero(kind[i:j] == RQT_ZERO)
                assert len(z) == 1 and np.array_equal(v[i][0:3], v[j][2:5])
                a, b = v[i], v[z[0]]
                rows.append([a[9], int(a[3]) + 4 * int(a[4]) + 4 * int(a[5]), i64(a[6:7], a[7:8])[0], int(b[0]) + 4 * int(b[1]) + 4 * int(b[2]), i64(b[3:4], b[4:5])[0], a[8]])
                chosen.append(depth)
        self.rows = np.array(rows, np.int64).reshape(-1, 6)
        self.chosen = np.array(chosen, np.int32)
        self.uncoded = uncoded

    def __len__(self):
        return len(self.rows)
