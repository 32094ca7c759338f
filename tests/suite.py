"""The parity suite: one seeded set of inputs + job tables covering every primitive of SURVEY.md 8(a), and a
runner that evaluates it through any implementation exposing the *batch* interface below:

    Oracle / Reference (per-call C functions, adapted by LoopImpl)   -> expected values
    turingcodec_amd.Havoc (libhavoc_mi355x.so, one launch per primitive) -> values under test

`make_inputs(seed)` returns a flat dict of numpy arrays (what tests/golden/havoc_golden.npz stores), `run(impl, d)`
returns a dict of output arrays keyed like the golden file.  Job tables are int32 arrays whose columns follow the
job structs of include/havoc_mi355x.h.
"""
import numpy as np

import cases
from cases import PLANE_W as W

SLOT = 64 * 64  # every block-producing job writes into its own 64x64 slot (stride 64) of the output plane


def _pairs(rng, sizes, n):
    return np.array([(ao, bo, w, h) for (w, h, ao, bo) in cases.block_pair_cases(rng, sizes, n)], np.int32)


def make_inputs(seed=20260927):
    rng = np.random.default_rng(seed)
    d = {}
    for S, bd in ((1, 8), (2, 10)):
        k = "u8" if S == 1 else "u16"
        d[f"{k}.a"] = cases.rand_plane(rng, S, bd).ravel()
        d[f"{k}.b"] = cases.rand_plane(rng, S, bd).ravel()
        d[f"{k}.hi"] = cases.rand_plane(rng, S, bd, kind="high").ravel()
        d[f"{k}.lo"] = cases.rand_plane(rng, S, bd, kind="low").ravel()
        d[f"{k}.x"] = cases.rand_plane(rng, S, bd, kind="extremes").ravel()
        d[f"{k}.sad.jobs"] = _pairs(rng, cases.PU_SIZES, 2)
        d[f"{k}.sad4.jobs"] = np.array([(so, *ros, w, h, 0) for (w, h, so, ros) in
                                        cases.sad4_cases(rng, cases.PU_SIZES, 2)], np.int32)
        d[f"{k}.ssd.jobs"] = _pairs(rng, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64)], 3)
        # full-pel SAD surfaces: every PU size, all candidates within +-8 of a random centre (and a 4-wide non-PU shape)
        d[f"{k}.surface.jobs"] = _pairs(rng, cases.PU_SIZES + [(4, 4), (20, 8)], 1)
        # SATD: single 2/4/8 Hadamards plus PU-sized jobs tiled as measureSatd does (incl. chroma sizes -> 4x4, 2x2)
        d[f"{k}.satd.jobs"] = _pairs(rng, [(2, 2), (4, 4), (8, 8)] + cases.PU_SIZES + [(6, 8), (2, 4), (12, 8)], 2)
        # one block against 1..16 candidate blocks (the sub-pel stage's candidates of one PU)
        mj = []
        for i, (w, h) in enumerate([(2, 2), (4, 4), (8, 8)] + cases.PU_SIZES + [(6, 8), (2, 4)]):
            cnt = 16 if i % 3 else 1 + (5 * i) % 16
            ao = cases.off(*cases.rand_pos(rng, w, h))
            bo = [cases.off(*cases.rand_pos(rng, w, h)) for _ in range(cnt)] + [0] * (16 - cnt)
            mj.append([ao, w, h, cnt] + bo)
        d[f"{k}.satdm.jobs"] = np.array(mj, np.int32)
        uni = cases.pred_uni_cases(rng, [bd])
        for taps in (8, 4):
            d[f"{k}.pred_uni{taps}.jobs"] = np.array(
                [(0, ro, w, h, xf, yf, 0, 0) for (t, w, h, xf, yf, _, ro) in uni if t == taps], np.int32)
            d[f"{k}.pred_uni{taps}.jobs"][:, 0] = np.arange(len(d[f"{k}.pred_uni{taps}.jobs"])) * SLOT
        bi = cases.pred_bi_cases(rng, [bd])
        for taps in (8, 4):
            j = np.array([(0, r0, r1, w, h, a, b_, c, e, 0, 0, 0) for (t, w, h, a, b_, c, e, _, r0, r1) in bi if t == taps],
                         np.int32)
            j[:, 0] = np.arange(len(j)) * SLOT
            d[f"{k}.pred_bi{taps}.jobs"] = j
        j = np.array([(0, po, so, w, h, 0, 0, 0) for (w, h, so, po) in cases.block_pair_cases(rng, cases.PU_SIZES, 1)],
                     np.int32)
        j[:, 0] = np.arange(len(j)) * SLOT
        d[f"{k}.subtract_bi.jobs"] = j
        # intra: neighbours for job i live at nb[i*160 + 80]
        ic = [(log2, mode, 1) for log2 in (2, 3, 4, 5) for mode in range(35)]
        ic += [(log2, mode, 0) for log2 in (2, 3, 4, 5) for mode in (1, 10, 26)]
        nbs = []
        jobs = []
        for i, (log2, mode, edge) in enumerate(ic):
            nb, c = cases.rand_neighbours(rng, S, bd, "extremes" if i % 7 == 3 else "uniform")
            nbs.append(nb)
            jobs.append((i * 1024, i * 160 + c, log2, mode, edge, 0, 0, 0))
        d[f"{k}.intra.nb"] = np.concatenate(nbs)
        d[f"{k}.intra.jobs"] = np.array(jobs, np.int32)
        # residual jobs: a = src (plane a), b = pred (plane b); output slots of 64x64 int16
        d[f"{k}.residual.jobs"] = _pairs(rng, [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (32, 16)], 1)
        # inverse transform + add: pred from plane a (stride W), dst slots stride 64
        tj = []
        co = 0
        coefs = []
        mx = (1 << bd) - 1
        for (log2, tr) in cases.TRANSFORMS:
            n = 1 << log2
            for rep, (lo, hi, kind) in enumerate([(-128, 127, "uniform"), (-mx, mx, "uniform"),
                                                  (-32768, 32767, "uniform"), (-32768, 32767, "extremes")]):
                x, y = cases.rand_pos(rng, n, n)
                tj.append((co, 0, cases.off(x, y), len(tj) * SLOT, log2, tr, 0, 0))
                coefs.append(cases.residual_block(rng, n, lo, hi, kind).ravel())
                co += n * n
        d[f"{k}.itx.coeffs"] = np.concatenate(coefs)
        d[f"{k}.itx.jobs"] = np.array(tj, np.int32)
    # forward transforms (bit depth 8 and 10): residual blocks in one int16 plane of stride 64, one 64x64 slot per job
    for bd in (8, 10):
        mx = (1 << bd) - 1
        tj = []
        res = []
        co = 0
        for (log2, tr) in cases.TRANSFORMS:
            n = 1 << log2
            for (lo, hi, kind) in [(-256, 255, "uniform"), (-mx, mx, "uniform"), (-mx, mx, "extremes"),
                                   (-32768, 32767, "uniform")]:
                res.append(cases.residual_block(rng, 64, lo, hi, kind).ravel())
                tj.append((co, len(tj) * SLOT + int(rng.integers(0, 16)) * 64 + int(rng.integers(0, 16)), 0, 0, log2, tr, 0, 0))
                co += n * n
        d[f"fwd{bd}.res"] = np.concatenate(res)
        d[f"fwd{bd}.jobs"] = np.array(tj, np.int32)
    # quantisers
    qsrc = np.concatenate([rng.integers(-32768, 32768, 4096), rng.integers(-300, 300, 4096)]).astype(np.int16)
    d["quant.src"] = qsrc
    qj = []
    dj = []
    for i, n in enumerate((16, 64, 256, 1024)):
        for base in (0, 4096):
            so = base + int(rng.integers(0, (4096 - n) // 16 + 1)) * 16
            for qp in (22, 27, 32, 37):
                for log2 in (2, 3, 4, 5):
                    for bd in (8, 10):
                        sc, sh, of = cases.quant_params(qp, log2, bd, (qp + log2) & 1)
                        if 16 <= sh <= 27 and len(qj) < 200:
                            qj.append((0, so, n, sc, sh, of, 0, 0))
                        sc, sh = cases.dequant_params(qp, log2, bd)
                        if int(np.abs(qsrc[so:so + n].astype(np.int64)).max()) * sc + (1 << (sh - 1)) < 2 ** 31 and len(dj) < 200:
                            dj.append((0, so, n, sc, sh, 0, 0, 0))
        qj.append((0, base, n, 51, 20, 14, 0, 0))          # havoc/quantize.cpp:523-525
        dj.append((0, 4096, n, 51, 1, 0, 0, 0))            # havoc/quantize.cpp:266-270
        dj.append((0, 4096, n, 52224, 1, 0, 0, 0))
    for tbl in (qj, dj):
        o = 0
        for r in range(len(tbl)):
            tbl[r] = (o,) + tbl[r][1:]
            o += tbl[r][2]
    d["quant.jobs"] = np.array(qj, np.int32)
    d["dequant.jobs"] = np.array(dj, np.int32)
    # quantize_reconstruct (8-bit): pred = plane a, res blocks contiguous
    rj = []
    rres = []
    co = 0
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        x, y = cases.rand_pos(rng, n, n)
        rj.append((co, co, cases.off(x, y), len(rj) * SLOT, log2, 0, 0, 0))
        rres.append(rng.integers(-300, 300, n * n).astype(np.int16))
        co += n * n
    d["qrec.res"] = np.concatenate(rres)
    d["qrec.jobs"] = np.array(rj, np.int32)
    d["ssd_linear.n"] = np.array([1, 16, 100, 512, 4096, 25600], np.int32)
    # fused 35-mode intra SATD stage: partitions of every size; source block in plane a, unfiltered + filtered
    # neighbour arrays (independent random data: the kernel must honour the per-mode filter mask), random masks
    for S, bd in ((1, 8), (2, 10)):
        k = "u8" if S == 1 else "u16"
        jobs, nbs = [], []
        for log2 in (2, 3, 4, 5):
            n = 1 << log2
            for rep in range(7 if log2 < 5 else 3):
                x, y = cases.rand_pos(rng, n, n)
                kind = "extremes" if rep == 2 else "uniform"
                nbu, c = cases.rand_neighbours(rng, S, bd, kind)
                nbf, _ = cases.rand_neighbours(rng, S, bd, kind)
                base = sum(len(a) for a in nbs)
                nbs += [nbu, nbf]
                mask = int(rng.integers(0, 1 << 35)) if rep else 0
                jobs.append((cases.off(x, y), base + c, base + len(nbu) + c, mask & 0xffffffff, mask >> 32, rep != 1, log2, 0))
        d[f"{k}.intra35.nb"] = np.concatenate(nbs)
        d[f"{k}.intra35.jobs"] = np.array(jobs, np.int64).astype(np.uint32).view(np.int32).reshape(len(jobs), 8)
    # fused TU chain: source in plane a, prediction in plane b, reconstruction slots (stride 64), small random levels
    for S in (1, 2):
        k = "u8" if S == 1 else "u16"
        tj, co = [], 0
        for (log2, tr) in cases.TRANSFORMS:
            n = 1 << log2
            for rep in range(5):
                sx, sy = cases.rand_pos(rng, n, n)
                px, py = cases.rand_pos(rng, n, n)
                tj.append((co, cases.off(sx, sy), cases.off(px, py), len(tj) * SLOT, log2, tr, 0, 0))
                co += n * n
        d[f"{k}.tuf.jobs"] = np.array(tj, np.int32)
        d[f"{k}.tuf.levels"] = rng.integers(-48, 49, co).astype(np.int16)
    # fused sub-pel candidates: the pred_uni job tables with dst_off replaced by a source-block offset in plane b
    for S in (1, 2):
        k = "u8" if S == 1 else "u16"
        for taps in (8, 4):
            j = d[f"{k}.pred_uni{taps}.jobs"].copy()
            j[:, 0] = [cases.off(*cases.rand_pos(rng, int(w), int(h))) for (w, h) in j[:, 2:4]]
            d[f"{k}.subpel{taps}.jobs"] = j
    # ---- round-2 additions.  APPENDED: every draw above is unchanged, so the earlier golden arrays stay what they were ----
    # (1) SAD / SAD4 at widths outside the 23 PU sizes -- the reference's sadGeneric entry (havoc/sad.h:77-89): 16-bit rows
    #     of 34..62 samples (row bytes % 8 == 4 and > 64), odd widths, a non-PU multiple of 4
    gen = [(34, 8), (38, 4), (62, 16), (46, 12), (7, 5), (13, 9), (33, 17), (63, 64), (20, 8), (5, 4)]
    for S in (1, 2):
        k = "u8" if S == 1 else "u16"
        d[f"{k}.sadg.jobs"] = _pairs(rng, gen, 2)
        d[f"{k}.sad4g.jobs"] = np.array([(so, *ros, w, h, 0) for (w, h, so, ros) in cases.sad4_cases(rng, gen, 1)], np.int32)
    # (2) 9-bit content on the 16-bit path: the reference's tables have a 9-bit row (havoc/pred_inter.h:40, pred_intra.h:37)
    d["u9.a"] = cases.rand_plane(rng, 2, 9).ravel()
    d["u9.b"] = cases.rand_plane(rng, 2, 9).ravel()
    d["u9.x"] = cases.rand_plane(rng, 2, 9, kind="extremes").ravel()
    uni = cases.pred_uni_cases(rng, [9])
    for taps in (8, 4):
        j = np.array([(0, ro, w, h, xf, yf, 0, 0) for (t, w, h, xf, yf, _, ro) in uni if t == taps], np.int32)
        j[:, 0] = np.arange(len(j)) * SLOT
        d[f"u9.pred_uni{taps}.jobs"] = j
        js = j.copy()
        js[:, 0] = [cases.off(*cases.rand_pos(rng, int(w), int(h))) for (w, h) in j[:, 2:4]]
        d[f"u9.subpel{taps}.jobs"] = js
    bi = cases.pred_bi_cases(rng, [9])
    for taps in (8, 4):
        j = np.array([(0, r0, r1, w, h, a, b_, c, e, 0, 0, 0) for (t, w, h, a, b_, c, e, _, r0, r1) in bi if t == taps], np.int32)
        j[:, 0] = np.arange(len(j)) * SLOT
        d[f"u9.pred_bi{taps}.jobs"] = j
    ic = [(log2, mode, 1) for log2 in (2, 3, 4, 5) for mode in range(35)] + [(log2, mode, 0) for log2 in (2, 3, 4, 5) for mode in (1, 10, 26)]
    nbs, jobs = [], []
    for i, (log2, mode, edge) in enumerate(ic):
        nb, c = cases.rand_neighbours(rng, 2, 9, "extremes" if i % 5 == 2 else "uniform")
        nbs.append(nb)
        jobs.append((i * 1024, i * 160 + c, log2, mode, edge, 0, 0, 0))
    d["u9.intra.nb"] = np.concatenate(nbs)
    d["u9.intra.jobs"] = np.array(jobs, np.int32)
    jobs, nbs = [], []
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        for rep in range(3):
            x, y = cases.rand_pos(rng, n, n)
            nbu, c = cases.rand_neighbours(rng, 2, 9, "extremes" if rep == 2 else "uniform")
            nbf, _ = cases.rand_neighbours(rng, 2, 9, "uniform")
            base = sum(len(a) for a in nbs)
            nbs += [nbu, nbf]
            mask = int(rng.integers(0, 1 << 35))
            jobs.append((cases.off(x, y), base + c, base + len(nbu) + c, mask & 0xffffffff, mask >> 32, rep != 1, log2, 0))
    d["u9.intra35.nb"] = np.concatenate(nbs)
    d["u9.intra35.jobs"] = np.array(jobs, np.int64).astype(np.uint32).view(np.int32).reshape(len(jobs), 8)
    tj, coefs, co = [], [], 0
    for (log2, tr) in cases.TRANSFORMS:
        n = 1 << log2
        for (lo, hi, kind) in [(-128, 127, "uniform"), (-511, 511, "uniform"), (-32768, 32767, "extremes")]:
            x, y = cases.rand_pos(rng, n, n)
            tj.append((co, 0, cases.off(x, y), len(tj) * SLOT, log2, tr, 0, 0))
            coefs.append(cases.residual_block(rng, n, lo, hi, kind).ravel())
            co += n * n
    d["u9.itx.coeffs"] = np.concatenate(coefs)
    d["u9.itx.jobs"] = np.array(tj, np.int32)
    return d


def run_round2(impl, d, want, out):
    """the round-2 additions of make_inputs (generic-width SAD, 9-bit on the 16-bit path)"""
    for S in (1, 2):
        k = "u8" if S == 1 else "u16"
        if want(f"{k}.sadg"):
            out[f"{k}.sadg"] = impl.sad(d[f"{k}.a"], W, d[f"{k}.b"], W, d[f"{k}.sadg.jobs"])
            out[f"{k}.sadg.x"] = impl.sad(d[f"{k}.x"], W, d[f"{k}.b"], W, d[f"{k}.sadg.jobs"])
            out[f"{k}.sad4g"] = impl.sad4(d[f"{k}.a"], W, d[f"{k}.b"], W, d[f"{k}.sad4g.jobs"])
    a, x = d["u9.a"], d["u9.x"]
    for taps in (8, 4):
        if want("u9.pred_uni"):
            j = d[f"u9.pred_uni{taps}.jobs"]
            out[f"u9.pred_uni{taps}"] = impl.pred_uni(taps, 9, len(j) * SLOT, 64, a, W, j)
            out[f"u9.pred_uni{taps}.x"] = impl.pred_uni(taps, 9, len(j) * SLOT, 64, x, W, j)
        if want("u9.pred_bi"):
            j = d[f"u9.pred_bi{taps}.jobs"]
            out[f"u9.pred_bi{taps}"] = impl.pred_bi(taps, 9, len(j) * SLOT, 64, a, W, j)
            out[f"u9.pred_bi{taps}.x"] = impl.pred_bi(taps, 9, len(j) * SLOT, 64, x, W, j)
        if want("u9.subpel"):
            out[f"u9.subpel{taps}"] = impl.subpel_satd(taps, 9, d["u9.b"], W, a, W, d[f"u9.subpel{taps}.jobs"])
    if want("u9.intra"):
        j = d["u9.intra.jobs"]
        out["u9.intra"] = impl.intra(9, len(j) * 1024, 32, d["u9.intra.nb"], j)
    if want("u9.intra35"):
        out["u9.intra35"] = impl.intra_satd35(9, a, W, d["u9.intra35.nb"], d["u9.intra35.jobs"])
    if want("u9.itx"):
        j = d["u9.itx.jobs"]
        out["u9.itx_add"] = impl.inverse_transform_add(9, len(j) * SLOT, 64, a, W, d["u9.itx.coeffs"], j)
    if want("u9.planes"):
        out["u9.planes"] = impl.interp_planes(9, a, W, 5, 6, 139, 141)


def run(impl, d, keys=None):
    """Evaluate the suite through `impl`; returns {name: ndarray}.  `keys`: optional substring filter."""
    out = {}

    def want(name):
        return keys is None or any(s in name for s in keys)

    for S, bd in ((1, 8), (2, 10)):
        k = "u8" if S == 1 else "u16"
        a, b = d[f"{k}.a"], d[f"{k}.b"]
        for nm, (p, q) in (("", (a, b)), (".edge", (d[f"{k}.hi"], d[f"{k}.lo"])), (".x", (d[f"{k}.x"], d[f"{k}.b"]))):
            if want(f"{k}.sad"):
                out[f"{k}.sad{nm}"] = impl.sad(p, W, q, W, d[f"{k}.sad.jobs"])
                out[f"{k}.sad4{nm}"] = impl.sad4(p, W, q, W, d[f"{k}.sad4.jobs"])
            if want(f"{k}.surface"):
                out[f"{k}.surface8{nm}"] = impl.sad_surface(p, W, q, W, 8, d[f"{k}.surface.jobs"])
            if want(f"{k}.ssd"):
                out[f"{k}.ssd{nm}"] = impl.ssd(p, W, q, W, d[f"{k}.ssd.jobs"])
            if want(f"{k}.satd"):
                out[f"{k}.satd{nm}"] = impl.satd(p, W, q, W, d[f"{k}.satd.jobs"])
                out[f"{k}.satdm{nm}"] = impl.satd_multi(p, W, q, W, d[f"{k}.satdm.jobs"])
        for taps in (8, 4):
            if want(f"{k}.pred_uni"):
                j = d[f"{k}.pred_uni{taps}.jobs"]
                out[f"{k}.pred_uni{taps}"] = impl.pred_uni(taps, bd, len(j) * SLOT, 64, a, W, j)
                out[f"{k}.pred_uni{taps}.x"] = impl.pred_uni(taps, bd, len(j) * SLOT, 64, d[f"{k}.x"], W, j)
            if want(f"{k}.pred_bi"):
                j = d[f"{k}.pred_bi{taps}.jobs"]
                out[f"{k}.pred_bi{taps}"] = impl.pred_bi(taps, bd, len(j) * SLOT, 64, a, W, j)
                out[f"{k}.pred_bi{taps}.x"] = impl.pred_bi(taps, bd, len(j) * SLOT, 64, d[f"{k}.x"], W, j)
        if want(f"{k}.subtract_bi"):
            j = d[f"{k}.subtract_bi.jobs"]
            out[f"{k}.subtract_bi"] = impl.subtract_bi(bd, len(j) * SLOT, 64, b, W, a, W, j)
            out[f"{k}.subtract_bi.x"] = impl.subtract_bi(bd, len(j) * SLOT, 64, d[f"{k}.x"], W, a, W, j)
        if want(f"{k}.intra"):
            j = d[f"{k}.intra.jobs"]
            out[f"{k}.intra"] = impl.intra(bd, len(j) * 1024, 32, d[f"{k}.intra.nb"], j)
        if want(f"{k}.residual"):
            j = d[f"{k}.residual.jobs"]
            out[f"{k}.residual"] = impl.residual(len(j) * SLOT, 64, np.arange(len(j), dtype=np.int32) * SLOT, a, W, b, W, j)
        if want(f"{k}.itx"):
            j = d[f"{k}.itx.jobs"]
            out[f"{k}.itx_add"] = impl.inverse_transform_add(bd, len(j) * SLOT, 64, a, W, d[f"{k}.itx.coeffs"], j)
            if S == 1:
                jr = j.copy()
                jr[:, 1] = j[:, 0]  # res_off = coef_off: n*n contiguous
                out["itx8"] = impl.inverse_transform(8, len(d[f"{k}.itx.coeffs"]), d[f"{k}.itx.coeffs"], jr)
            else:
                jr = j.copy()
                jr[:, 1] = j[:, 0]
                out["itx10"] = impl.inverse_transform(10, len(d[f"{k}.itx.coeffs"]), d[f"{k}.itx.coeffs"], jr)
    for bd in (8, 10):
        if want(f"fwd{bd}"):
            j = d[f"fwd{bd}.jobs"]
            ncoef = int(sum((1 << (2 * int(r[4]))) for r in j))
            out[f"fwd{bd}"] = impl.transform(bd, ncoef, d[f"fwd{bd}.res"], 64, j)
    if want("quant"):
        j = d["quant.jobs"]
        q, cbf = impl.quantize(int(j[:, 2].sum()), d["quant.src"], j)
        out["quant"] = q
        out["quant.cbf"] = (np.asarray(cbf) != 0).astype(np.int32)
        j = d["dequant.jobs"]
        out["dequant"] = impl.quantize_inverse(int(j[:, 2].sum()), d["quant.src"], j)
    if want("qrec"):
        j = d["qrec.jobs"]
        out["qrec"] = impl.quantize_reconstruct(len(j) * SLOT, 64, d["u8.a"], W, d["qrec.res"], j)
    if want("ssd_linear"):
        out["ssd_linear"] = np.array([impl.ssd_linear(d["u8.a"], d["u8.b"], int(n)) for n in d["ssd_linear.n"]], np.int64)
    for S, bd in ((1, 8), (2, 10)):
        k = "u8" if S == 1 else "u16"
        if want(f"{k}.intra35"):
            out[f"{k}.intra35"] = impl.intra_satd35(bd, d[f"{k}.a"], W, d[f"{k}.intra35.nb"], d[f"{k}.intra35.jobs"])
        if want(f"{k}.tuf"):
            j = d[f"{k}.tuf.jobs"]
            out[f"{k}.tuf.coef"] = impl.tu_forward(bd, len(d[f"{k}.tuf.levels"]), d[f"{k}.a"], W, d[f"{k}.b"], W, j)
            rec, ssd = impl.tu_reconstruct(bd, 32, len(j) * SLOT, 64, d[f"{k}.b"], W, d[f"{k}.a"], W, d[f"{k}.tuf.levels"], j)
            out[f"{k}.tuf.rec"], out[f"{k}.tuf.ssd"] = rec, np.asarray(ssd, np.uint32)
        if want(f"{k}.planes"):
            # a rectangle that is not a multiple of the 64x16 kernel tile, inside the 160x160 plane with the required margin
            out[f"{k}.planes"] = impl.interp_planes(bd, d[f"{k}.a"], W, 5, 6, 139, 141)
            out[f"{k}.planes.x"] = impl.interp_planes(bd, d[f"{k}.x"], W, 16, 16, 64, 16)
        for taps in (8, 4):
            if want(f"{k}.subpel"):
                out[f"{k}.subpel{taps}"] = impl.subpel_satd(taps, bd, d[f"{k}.b"], W, d[f"{k}.a"], W, d[f"{k}.subpel{taps}.jobs"])
                out[f"{k}.subpel{taps}.x"] = impl.subpel_satd(taps, bd, d[f"{k}.b"], W, d[f"{k}.x"], W, d[f"{k}.subpel{taps}.jobs"])
    if "u9.a" in d:
        run_round2(impl, d, want, out)
    return out


class LoopImpl:
    """Batch interface on top of a per-call checker (reflibs.Oracle or reflibs.Reference)."""

    def __init__(self, f, pu_satd=None):
        self.f = f
        self._pu_satd = pu_satd

    def sad(self, a, sa, b, sb, jobs):
        return np.array([self.f.sad(a, int(j[0]), sa, b, int(j[1]), sb, int(j[2]), int(j[3])) for j in jobs], np.int32)

    def sad4(self, a, sa, b, sb, jobs):
        return np.array([self.f.sad4(a, int(j[0]), sa, b, [int(x) for x in j[1:5]], sb, int(j[5]), int(j[6]))
                         for j in jobs], np.int32)

    def sad_surface(self, a, sa, b, sb, rng, jobs):
        """the surface is by definition the single SAD of every candidate (havoc/sad.cpp:432-449)"""
        side = 2 * rng + 1
        out = np.zeros((len(jobs), side, side), np.int32)
        for i, j in enumerate(jobs):
            for dy in range(-rng, rng + 1):
                for dx in range(-rng, rng + 1):
                    out[i, dy + rng, dx + rng] = self.f.sad(a, int(j[0]), sa, b, int(j[1]) + dy * sb + dx, sb, int(j[2]), int(j[3]))
        return out

    def ssd(self, a, sa, b, sb, jobs):
        return np.array([self.f.ssd(a, int(j[0]), sa, b, int(j[1]), sb, int(j[2]), int(j[3])) for j in jobs], np.uint32)

    def satd_multi(self, a, sa, b, sb, jobs):
        """by definition the PU SATD of every (a, b_k) pair"""
        out = np.zeros((len(jobs), 16), np.int32)
        for i, j in enumerate(jobs):
            pairs = np.array([(j[0], j[4 + k], j[1], j[2]) for k in range(int(j[3]))], np.int32)
            out[i, :int(j[3])] = self.satd(a, sa, b, sb, pairs)
        return out

    def satd(self, a, sa, b, sb, jobs):
        res = []
        for j in jobs:
            ao, bo, w, h = (int(x) for x in j)
            n = 2 if (w | h) & 3 else (4 if (w | h) & 7 else 8)   # turing/Measure.h:100-134
            t = 0
            for y in range(0, h, n):
                for x in range(0, w, n):
                    t += self.f.satd(a, ao + y * sa + x, sa, b, bo + y * sb + x, sb, n)
            res.append(t)
        return np.array(res, np.int32)

    def ssd_linear(self, a, b, n):
        return self.f.ssd_linear(a, b, n)

    def pred_uni(self, taps, bd, dst_len, sd, ref, sr, jobs):
        dst = np.zeros(dst_len, ref.dtype)
        for j in jobs:
            do, ro, w, h, xf, yf = (int(x) for x in j[:6])
            self.f.pred_uni(dst, do, sd, ref, ro, sr, w, h, xf, yf, bd, taps)
            self._trim(dst, do, sd, w, h)
        return dst

    @staticmethod
    def _trim(dst, do, sd, w, h):
        """zero whatever a SIMD implementation wrote to the right of the block (havoc/pred_inter.h:27)"""
        blk = dst[do:do + SLOT].reshape(64, 64) if sd == 64 else None
        if blk is not None:
            blk[:h, w:] = 0
            blk[h:, :] = 0

    def pred_bi(self, taps, bd, dst_len, sd, ref, sr, jobs):
        dst = np.zeros(dst_len + 256, ref.dtype)
        for j in jobs:
            do, r0, r1, w, h, a, b, c, e = (int(x) for x in j[:9])
            self.f.pred_bi(dst, do, sd, ref, r0, r1, sr, w, h, a, b, c, e, bd, taps)
            self._trim(dst, do, sd, w, h)
        return dst[:dst_len]

    def subtract_bi(self, bd, dst_len, sd, pred, sp, src, ss, jobs):
        dst = np.zeros(dst_len, src.dtype)
        for j in jobs:
            do, po, so, w, h = (int(x) for x in j[:5])
            self.f.subtract_bi(dst, do, sd, pred, po, sp, src, so, ss, w, h, bd)
        return dst

    def intra(self, bd, dst_len, sd, nb, jobs):
        dst = np.zeros(dst_len, nb.dtype)
        for j in jobs:
            do, no, log2, mode, edge = (int(x) for x in j[:5])
            if hasattr(self.f, "L") and hasattr(self.f.L, "oracle_intra"):
                self.f.intra(dst, do, sd, nb, no, log2, mode, 1 if (edge and log2 < 5) else 0, bd)
            else:
                self.f.intra(dst, do, sd, nb, no, log2, mode, edge, bd)
        return dst

    def tu_forward(self, bd, ncoef, src, ss, pred, sp, jobs):
        """residual loop (turing/Reconstruct.cpp:258-260) then the forward transform, per TU"""
        co = np.zeros(ncoef, np.int16)
        for j in np.asarray(jobs, np.int32):
            c, so, po, _, log2, tr = (int(v) for v in j[:6])
            n = 1 << log2
            res = self.residual(n * n, n, [0], src, ss, pred, sp, np.array([[so, po, n, n]], np.int32))
            self.f.transform(co, c, res, 0, n, log2, tr, bd)
        return co

    def tu_reconstruct(self, bd, qp, rec_len, sr, pred, sp, src, ss, levels, jobs):
        """havoc_quantize_inverse -> inverse_transform_add -> havoc_ssd(src, rec), per TU (Reconstruct.cpp:315-353)"""
        rec = np.zeros(rec_len, src.dtype)
        ssd = []
        for j in np.asarray(jobs, np.int32):
            c, so, po, ro, log2, tr = (int(v) for v in j[:6])
            n = 1 << log2
            scale, shift = cases.dequant_params(qp, log2, bd)
            dq = np.zeros(n * n, np.int16)
            self.f.quantize_inverse(dq, 0, levels, c, scale, shift, n * n)
            self.f.inverse_transform_add(rec, ro, sr, pred, po, sp, dq, 0, log2, tr, bd)
            ssd.append(self.f.ssd(src, so, ss, rec, ro, sr, n, n))
        return rec, np.array(ssd, np.uint32)

    def interp_planes(self, bd, ref, stride, x0, y0, width, height):
        """every sample of plane (xf, yf) = what pred_uni writes there when the sample is part of any block: evaluate
        pred_uni on <= 64x64 blocks tiling the rectangle"""
        planes = np.zeros((16, len(ref)), ref.dtype)
        for yf in range(4):
            for xf in range(4):
                if xf == 0 and yf == 0:
                    continue
                for by in range(y0, y0 + height, 64):
                    for bx in range(x0, x0 + width, 64):
                        w, h = min(64, x0 + width - bx), min(64, y0 + height - by)
                        self.f.pred_uni(planes[4 * yf + xf], by * stride + bx, stride, ref, by * stride + bx, stride, w, h, xf, yf, bd, 8)
        return planes

    def subpel_satd(self, taps, bd, src, ss, ref, sr, jobs):
        """costDistortionMv (turing/Search.hpp:1965-1998): pred_uni into a 64-stride scratch block, then measureSatd"""
        out = np.zeros(len(jobs), np.int32)
        for i, j in enumerate(np.asarray(jobs, np.int32)):
            so, ro, w, h, xf, yf = (int(v) for v in j[:6])
            one = np.array([[0, ro, w, h, xf, yf, 0, 0]], np.int32)
            pred = self.pred_uni(taps, bd, SLOT, 64, ref, sr, one)
            out[i] = self.satd(src, ss, pred, 64, np.array([[so, 0, w, h]], np.int32))[0]
        return out

    def intra_satd35(self, bd, src, ss, nb, jobs):
        """composition of the two per-call primitives, as PredictIntraLumaBlock does (turing/Reconstruct.cpp:630-701):
        predict into a 32-stride scratch block, then SATD against the source in 8x8 tiles (one 4x4 for 4x4 blocks)"""
        out = np.zeros((len(jobs), 35), np.int32)
        pred = np.zeros(32 * 32, nb.dtype)
        for i, j in enumerate(np.asarray(jobs, np.int64)):
            so, no, nfo, lo, hi, edge, log2 = (int(v) for v in j[:7])
            mask = (lo & 0xffffffff) | ((hi & 0xffffffff) << 32)
            n = 1 << log2
            for mode in range(35):
                sel = nfo if (mask >> mode) & 1 else no
                one = np.array([[0, sel, log2, mode, 1 if edge else 0, 0, 0, 0]], np.int32)
                p = self.intra(bd, 32 * 32, 32, nb, one)
                pred[:] = p
                t = n if n < 8 else 8
                c = 0
                for y in range(0, n, t):
                    for x in range(0, n, t):
                        c += self.f.satd(src, so + y * ss + x, ss, pred, y * 32 + x, 32, t)
                out[i, mode] = c
        return out

    def residual(self, res_len, sres, res_off, src, ss, pred, sp, jobs):
        res = np.zeros(res_len, np.int16)
        for ro, j in zip(res_off, jobs):
            so, po, w, h = (int(x) for x in j)
            if hasattr(self.f, "residual"):
                self.f.residual(res, int(ro), sres, src, so, ss, pred, po, sp, w, h)
            else:  # the reference has no such function (inline loops in turing/Reconstruct.cpp:258-260)
                for y in range(h):
                    res[int(ro) + y * sres:int(ro) + y * sres + w] = (
                        src[so + y * ss:so + y * ss + w].astype(np.int32) - pred[po + y * sp:po + y * sp + w].astype(np.int32))
        return res

    def transform(self, bd, ncoef, res, stride, jobs):
        co = np.zeros(ncoef, np.int16)
        for j in jobs:
            self.f.transform(co, int(j[0]), res, int(j[1]), stride, int(j[4]), int(j[5]), bd)
        return co

    def inverse_transform(self, bd, nres, coeffs, jobs):
        res = np.zeros(nres, np.int16)
        for j in jobs:
            self.f.inverse_transform(res, int(j[1]), coeffs, int(j[0]), int(j[4]), int(j[5]), bd)
        return res

    def inverse_transform_add(self, bd, dst_len, sd, pred, sp, coeffs, jobs):
        dst = np.zeros(dst_len, pred.dtype)
        for j in jobs:
            self.f.inverse_transform_add(dst, int(j[3]), sd, pred, int(j[2]), sp, coeffs, int(j[0]), int(j[4]), int(j[5]), bd)
        return dst

    def quantize(self, nout, src, jobs):
        dst = np.zeros(nout, np.int16)
        cbf = [self.f.quantize(dst, int(j[0]), src, int(j[1]), int(j[3]), int(j[4]), int(j[5]), int(j[2])) for j in jobs]
        return dst, np.array(cbf, np.int32)

    def quantize_inverse(self, nout, src, jobs):
        dst = np.zeros(nout, np.int16)
        for j in jobs:
            self.f.quantize_inverse(dst, int(j[0]), src, int(j[1]), int(j[3]), int(j[4]), int(j[2]))
        return dst

    def quantize_reconstruct(self, dst_len, sr, pred, sp, res, jobs):
        rec = np.zeros(dst_len, np.uint8)
        for j in jobs:
            self.f.quantize_reconstruct(rec, int(j[3]), sr, pred, int(j[2]), sp, res, int(j[1]), 1 << int(j[4]))
        return rec
