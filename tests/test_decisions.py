"""The decision-driven picture step (turingcodec_amd.decisions.DecisionPicture = what `bench.py --decisions` times): motion searches in
wavefront order with predictors derived from earlier decisions, then the TU chain on the chosen vectors -- against the same step composed
from the reference's own compiled functions on the host (oracle/_ref: havoc tables, Rdoq.cpp) and the sequential per-call walk.
Vectors, costs, the final motion field, coefficients, RDOQ levels, coded-block flags, SSDs and the reconstruction must agree bit for bit."""
import os

import numpy as np
import pytest

import reflibs
import search_tools as st

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(reflibs.REF_SO), reason="oracle/_ref not built")]


def _host_chain(R, dp, field):
    """the TU chain of DecisionPicture.tu_chain_fixed through the reference's functions, block by block"""
    from turingcodec_amd.workload import dequant_params, picture_lambda, quant_params
    W, PAD, stride, BD = dp.W, dp.PAD, dp.stride, dp.bd
    src, ref0 = dp.host_planes[0], dp.host_planes[1]
    pred = np.zeros(W * dp.H, src.dtype)
    recon = np.zeros(dp.pe, src.dtype)
    lam = picture_lambda(dp.qp)
    out = []
    for g in dp.groups:
        log2, N, m = g["log2"], g["nn"], g["m"]
        qs, qshift, _ = quant_params(dp.qp, log2, BD, False)
        inv, dshift = dequant_params(dp.qp, log2, BD)
        coef = np.zeros(m * N * N, np.int16)
        level, deq = np.zeros_like(coef), np.zeros_like(coef)
        cbf, ssd = np.zeros(m, np.int32), np.zeros(m, np.uint32)
        rows = np.arange(N)[:, None]
        for i in range(m):
            x0, y0 = int(g["x0"][i]), int(g["y0"][i])
            qx, qy = (int(v) for v in field[0, y0 >> 2, x0 >> 2])
            so = (y0 + PAD) * stride + x0 + PAD
            R.pred_uni(pred, y0 * W + x0, W, ref0, (y0 + (qy >> 2) + PAD) * stride + x0 + (qx >> 2) + PAD, stride, N, N, qx & 3, qy & 3, BD, 8)
            r16 = src[so + rows * stride + np.arange(N)].astype(np.int16) - pred[y0 * W + x0 + rows * W + np.arange(N)].astype(np.int16)
            R.transform(coef, i * N * N, np.ascontiguousarray(r16).ravel(), 0, N, log2, 0, BD)
            ctu = (y0 // 64) * dp.cx + x0 // 64
            lv, c = R.rdoq(np.ascontiguousarray(coef[i * N * N:(i + 1) * N * N]), log2, 0, 0, 0, 1, qs, qshift, inv, BD, lam, dp.rdoq_states[ctu])
            level[i * N * N:(i + 1) * N * N] = lv
            cbf[i] = c
            R.quantize_inverse(deq, i * N * N, level, i * N * N, inv, dshift, N * N)
            R.inverse_transform_add(recon, so, stride, pred, y0 * W + x0, W, deq, i * N * N, log2, 0, BD)
            ssd[i] = R.ssd(src, so, stride, recon, so, stride, N, N)
        out.append(dict(coef=coef, level=level, cbf=cbf, ssd=ssd))
    return out, recon


def _pu_satd(R, a, ao, sa, b, bo, sb, w, h):
    """measureSatd (turing/Measure.h:97-135) through the reference's tile functions: 8x8 tiles where both sides are multiples of 8, else 4x4"""
    n = 8 if (w | h) % 8 == 0 else 4
    return sum(R.satd(a, ao + y * sa + x, sa, b, bo + y * sb + x, sb, n) for y in range(0, h, n) for x in range(0, w, n))


def _check_merge_and_chroma(R, dp, field):
    """round 4: the merge candidates' three-plane SATD costs and the chroma TU chain of the step against the reference's own
    HavocPredBi / HavocPredUni (8- and 4-tap), Hadamard tiles, transform tables and Rdoq.cpp, unit by unit on the host"""
    from turingcodec_amd.workload import dequant_params, quant_params
    hv, BD, K, PAD, CP = dp.hv, dp.bd, dp.MERGE_CANDIDATES, dp.PAD, dp.PAD // 2
    u, mg = dp.units, dp.merge
    assert np.array_equal(mg["vectors"], dp.merge_vectors(field)) and (mg["vectors"] != 0).any()
    luma = np.concatenate([np.pad(p, (0, dp.pe - dp.n)) for p in dp.host_planes])
    chroma = np.concatenate([np.pad(p, (0, dp.cpe - dp.cn)) for p in dp.host_chroma])
    lam_q16 = int(float(dp.params.reciprocal_sqrt_lambda) * 65536 + 0.5)
    checked = 0
    for i in range(0, len(u), max(1, len(u) // 80)):      # a sample of the units, all five candidates, all three planes
        x0, y0, nn = int(u["x0"][i]), int(u["y0"][i]), 1 << int(u["log2_size"][i])
        costs = []
        for k in range(K):
            (ax, ay), (bx, by) = (int(v) for v in mg["vectors"][i, k, 0]), (int(v) for v in mg["vectors"][i, k, 1])
            want = []
            for plane in range(3):
                c = plane > 0
                size, stride, pad, pe, buf = (nn // 2, dp.cstride, CP, dp.cpe, chroma) if c else (nn, dp.stride, PAD, dp.pe, luma)
                px, py, sh, fm, taps = (x0 // 2, y0 // 2, 3, 7, 4) if c else (x0, y0, 2, 3, 8)
                first = 3 * (plane - 1) if c else 0
                at = lambda k_, mx, my: (first + k_) * pe + (py + (my >> sh) + pad) * stride + px + (mx >> sh) + pad
                pred = np.zeros(size * size, buf.dtype)
                R.pred_bi(pred, 0, size, buf, at(1, ax, ay), at(2, bx, by), stride, size, size, ax & fm, ay & fm, bx & fm, by & fm, BD, taps)
                want.append(_pu_satd(R, buf, first * pe + (py + pad) * stride + px + pad, stride, pred, 0, size, size, size))
            assert list(mg["satd"][i, k]) == want, (i, k, list(mg["satd"][i, k]), want)
            costs.append((min(k + 1, 4) << 16) + sum(want) * lam_q16)
        assert list(mg["cost"][i]) == costs and int(mg["best"][i]) == int(np.argmin(costs)), i
        checked += 1
    assert checked >= 40 and len(np.unique(mg["best"])) > 1
    # the chroma chain: every unit's Cb and Cr block
    hw, hh = dp.W // 2, dp.H // 2
    cpred, crecon = np.zeros(2 * hw * hh, luma.dtype), np.zeros(2 * dp.cpe, luma.dtype)
    coded = 0
    for g in dp.cgroups:
        cl, cn, comp, m = g["log2"], g["cn"], g["comp"], g["m"]
        qs, qshift, _ = quant_params(dp.qp, cl, BD, False)
        inv, dshift = dequant_params(dp.qp, cl, BD)
        coef, level, deq = np.zeros(m * cn * cn, np.int16), np.zeros(m * cn * cn, np.int16), np.zeros(m * cn * cn, np.int16)
        cbf, ssd = np.zeros(m, np.int32), np.zeros(m, np.uint32)
        rows = np.arange(cn)[:, None]
        for j in range(m):
            ui = g["sel"][j]
            qx, qy = (int(v) for v in field[0, int(u["y0"][ui]) >> 2, int(u["x0"][ui]) >> 2])
            x0, y0 = int(u["x0"][ui]) // 2, int(u["y0"][ui]) // 2
            so = 3 * (comp - 1) * dp.cpe + (y0 + CP) * dp.cstride + x0 + CP
            po = (comp - 1) * hw * hh + y0 * hw + x0
            ro = (comp - 1) * dp.cpe + (y0 + CP) * dp.cstride + x0 + CP
            R.pred_uni(cpred, po, hw, chroma, (3 * (comp - 1) + 1) * dp.cpe + (y0 + (qy >> 3) + CP) * dp.cstride + x0 + (qx >> 3) + CP, dp.cstride, cn, cn, qx & 7, qy & 7, BD, 4)
            r16 = chroma[so + rows * dp.cstride + np.arange(cn)].astype(np.int16) - cpred[po + rows * hw + np.arange(cn)].astype(np.int16)
            R.transform(coef, j * cn * cn, np.ascontiguousarray(r16).ravel(), 0, cn, cl, 0, BD)
            ctu = (int(u["y0"][ui]) // 64) * dp.cx + int(u["x0"][ui]) // 64
            lv, c = R.rdoq(np.ascontiguousarray(coef[j * cn * cn:(j + 1) * cn * cn]), cl, comp, 0, 0, 1, qs, qshift, inv, BD, dp.lam, dp.rdoq_states[ctu])
            level[j * cn * cn:(j + 1) * cn * cn] = lv
            cbf[j] = c
            R.quantize_inverse(deq, j * cn * cn, level, j * cn * cn, inv, dshift, cn * cn)
            R.inverse_transform_add(crecon, ro, dp.cstride, cpred, po, hw, deq, j * cn * cn, cl, 0, BD)
            ssd[j] = R.ssd(chroma, so, dp.cstride, crecon, ro, dp.cstride, cn, cn)
        for name, want in (("coef", coef), ("level", level), ("cbf", cbf), ("ssd", ssd)):
            assert np.array_equal(hv.down(g[name], want.dtype), want), (cl, comp, name)
        coded += int((cbf != 0).sum())
    assert np.array_equal(hv.down(dp.cpred, luma.dtype), cpred) and np.array_equal(hv.down(dp.crecon, luma.dtype), crecon)
    return checked, coded


@pytest.mark.parametrize("res,BD,qp", [((416, 240), 8, 32), ((640, 360), 10, 27), ((640, 360), 8, 22)])
def test_decision_step_equals_the_reference_functions(res, BD, qp):
    from turingcodec_amd.decisions import DecisionPicture
    from turingcodec_amd.havoc import Havoc
    hv = Havoc(stream="new")
    dp = DecisionPicture(hv, res[0], res[1], BD, qp, seed=21, threads=8)
    dp.step()
    dp.step()                        # the second step records the fixed launch sequences after the searches into HIP graphs, the third replays them
    got, field, stats = dp.step()
    assert all(dp._graphs.values()) and len(dp._graphs) == 1      # everything after the searches: one graph
    ref = st.Client("ref", 3)
    exp, exp_field, exp_bi = ref.picture_uni(dp.params, dp.host_planes[0], dp.host_planes[1], dp.host_planes[2], dp.stride, dp.PAD, dp.pus, dp.ctu_first, dp.cx, dp.cy,
                                             dp.mvp_rate, bi=True)
    for k in ("mv", "mvd", "mv_integer", "mvp_flag", "wrote_2Nx2N", "calls", "cost_integer", "cost_subpel", "cost_mvd_zero"):
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(field, exp_field)
    for k in ("mv", "mvd", "mvp_flag", "calls", "cost_subpel"):      # the bi-directional refinements (searchBi) of the step
        assert np.array_equal(dp.bi_results[k], exp_bi[k]), k
    assert (dp.bi_results["calls"] > 0).mean() > 0.9
    assert stats.launches < len(got) and stats.steps == dp.cx + 2 * (dp.cy - 1)
    # the merge candidates (three planes, bi-predictive) and the chroma TU chain of the step against the reference's functions
    checked, coded = _check_merge_and_chroma(reflibs.Reference(), dp, exp_field)
    print(res, BD, qp, "merge units checked", checked, "coded chroma blocks", coded)
    # the residual-quadtree decisions of the step (both depths of every unit in one chain per transform size) against the same decisions
    # taken one block at a time through the reference's tables + Rdoq.cpp, on the prediction the device made from the decided vectors
    pred = hv.down(dp.pred, dp.dt).copy()
    exp_rqt, exp_rec = ref.rqt(BD, dp.host_planes[0], dp.stride, dp.PAD, pred, dp.W, dp.rdoq_states, dp.quant, dp.lam, 1.0 / dp.lam, dp.units)
    assert dp.rqt_results.tobytes() == exp_rqt.tobytes()
    assert dp.rqt_stats.launches <= 5 * 4 + 4 and (exp_rqt["depth"] == 1).any() and (exp_rqt["depth"] == 0).any()
    # ... the loop filter that follows in the step: strengths derived on the device from the decided block structure == the reference's own
    # derivation (LoopFilter.h processCu / Tu / Rc + sameMotion over the same units), then its deblocking templates and padding
    R = reflibs.Reference()
    u = dp.units
    size = 1 << u["log2_size"]
    mv0 = exp_field[0, u["y0"] >> 2, u["x0"] >> 2]
    cus = np.stack([u["x0"], u["y0"], u["log2_size"], np.zeros_like(size), np.full_like(size, qp), np.zeros_like(size)], 1)
    pus = np.stack([u["x0"], u["y0"], size, size, mv0[:, 0], mv0[:, 1], np.zeros_like(size), np.zeros_like(size), np.zeros_like(size), np.full_like(size, -1)], 1)
    tus = []
    for i in range(len(u)):
        if exp_rqt["depth"][i] == 1:
            h = int(size[i]) // 2
            tus += [(int(u["x0"][i]) + (k & 1) * h, int(u["y0"][i]) + (k >> 1) * h, int(u["log2_size"][i]) - 1, int(exp_rqt["one"]["cbf"][i, k] != 0), 0) for k in range(4)]
        else:
            tus.append((int(u["x0"][i]), int(u["y0"][i]), int(u["log2_size"][i]), int(exp_rqt["tried_zero"][i] == 1 and exp_rqt["zero"]["cbf"][i] != 0), 0))
    want_data, want_bs = R.derive_bs(dp.W, dp.H, cus, pus, np.array(tus, np.int32))
    assert np.array_equal(hv.down(dp.d_data, np.int8), want_data) and np.array_equal(hv.down(dp.d_bs, np.uint8), want_bs)
    assert (want_bs != 0).any()
    cb = np.full((dp.H // 2) * (dp.W // 2), 128 << (BD - 8), dp.dt)
    cr = cb.copy()
    o = dp.PAD * dp.stride + dp.PAD
    R.deblock(exp_rec[o:], dp.stride, cb, cr, dp.W // 2, dp.W, dp.H, BD, want_data, want_bs)
    R.pad_block(exp_rec, o, dp.W, dp.H, dp.stride, dp.PAD, True, True, True, True)
    assert np.array_equal(hv.down(dp.recon, dp.dt)[:dp.n], exp_rec)
    # ... the intra side of the step: 35-mode stage + refinement order, then the RD refinement of every candidate, against the per-call loops
    # through the reference's intra table, transform tables and Rdoq.cpp
    assert set(dp.intra_results) == set(dp.intra_parts) and len(dp.intra_parts) == 4
    on_device = dict(dp.intra_results)                      # what the step left: decisions taken by kernels between the launches
    on_device_rec = {log2: hv.down(g["d_rec"], dp.dt) for log2, g in dp.intra_parts.items()}
    on_host = dp.intra_decisions(on_device=False)           # the two-call route: decisions on the host
    for log2, (order, best) in on_host.items():
        g = dp.intra_parts[log2]
        exp_order = ref.intra_order(g["ictx"], dp.rsl, ref.intra35(BD, log2, dp.host_planes[0], dp.stride, g["nb"], g["jobs"]))
        assert order.tobytes() == exp_order.tobytes(), log2
        exp_best, exp_irec = ref.intra_rd(BD, log2, dp.host_planes[0], dp.stride, g["nb"], g["jobs"], exp_order, g["ictx"], g["ctu"], dp.rdoq_states, dp.quant[log2 - 2],
                                          dp.lam, 1.0 / dp.lam)
        assert best.tobytes() == exp_best.tobytes(), log2
        assert on_device[log2][1].tobytes() == exp_best.tobytes(), log2
        assert np.array_equal(hv.down(g["d_rec"], dp.dt).reshape(exp_irec.shape), exp_irec), log2
        assert np.array_equal(on_device_rec[log2].reshape(exp_irec.shape), exp_irec), log2
    # the fixed-size chain (16x16 blocks) on the same vectors: every intermediate against the reference's functions
    import torch
    with torch.cuda.stream(hv.tstream):
        dp.recon.zero_()      # the loop filter above left a filtered, padded picture there
    dp.tu_chain_fixed(field)
    hv.sync()
    dev, dev_recon = dp.results()
    host, host_recon = _host_chain(reflibs.Reference(), dp, exp_field)
    for d, h in zip(dev, host):
        for k in ("coef", "level", "cbf", "ssd"):
            assert np.array_equal(d[k], h[k]), (d["log2"], k)
    assert np.array_equal(dev_recon, host_recon)
    # the step did something: vectors moved off zero, coded and uncoded blocks
    assert (got["mv"] != 0).any() and any((d["cbf"] != 0).any() for d in dev)
    hv.close()
