"""The path and the steps either side of it, chained on the device, against the same chain composed from the reference's own
compiled functions on the host -- a toy inter encoder for two pictures (SURVEY.md 8(f) 1-4 in concert):

  picture store in HBM (source, reference, reconstruction)          [8(f)-4]
  -> motion search of every 16x16 block, decision loops fed by batch launches (libhavoc_search.so)   [8(f)-1]
  -> HavocPredUni at the chosen vectors -> residual + forward DCT -> Rdoq::runQuantisation           [a5, a9, a13, 8(f)-2]
  -> de-quantise + inverse DCT + add -> SSD                                                          [a11, a12, a3]
  -> in-loop deblocking (boundary strengths from the vectors and coded-block flags)
  -> sample-adaptive offset (statistics per CTU, a choice, band / edge filter) -> padding            [8(f)-3]
  -> the reconstruction becomes the reference the NEXT picture's motion search reads (phase planes interpolated from it).

Host twin: the same steps through oracle/_ref (the reference's havoc tables, Rdoq.cpp, LoopFilter.h, Padding.h) and the
restated decision loops over those tables.  Everything must agree bit for bit: vectors and costs, coefficients, levels,
coded-block flags, SSDs, the deblocked + padded reconstruction, and the second picture's vectors found in it.
"""
import ctypes as C
import os

import numpy as np
import pytest

import reflibs
import search_tools as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(reflibs.REF_SO), reason="oracle/_ref not built")]

W, H, PAD, QP, N = 416, 240, 96, 32, 16     # picture, border, slice QP, block size


class Stats(C.Structure):
    _fields_ = [("rounds", C.c_int32), ("launches", C.c_int32), ("surfaces_small", C.c_int32), ("surfaces_large", C.c_int32),
                ("satd_jobs", C.c_int32), ("replays", C.c_int32), ("bytes_down", C.c_int64), ("seconds_gpu", C.c_double),
                ("seconds_host", C.c_double), ("seconds_total", C.c_double)]


def _blocks(mvp):
    """one 2Nx2N search per 16x16 block, list 0, predictors = `mvp` (quarter samples) and zero"""
    xs, ys = np.meshgrid(np.arange(0, W, N), np.arange(0, H, N))
    pus = np.zeros(xs.size, st.PU_DT)
    pus["x0"], pus["y0"], pus["w"], pus["h"] = xs.ravel(), ys.ravel(), N, N
    pus["cu_log2_size"], pus["cqt_depth"], pus["part_2Nx2N"] = 4, 2, 1
    pus["x_ctb"], pus["y_ctb"] = pus["x0"] & ~63, pus["y0"] & ~63
    pus["mvp"][:, 0] = mvp
    pus["mvp_rate"] = (40000, 90000)
    return pus


def _strengths(mv, cbf):
    """LoopFilter::Block arrays for a picture of 16x16 inter blocks: strength 1 on a block edge when either side has coefficients
    or the vectors differ by a sample or more (HEVC 8.7.2.4), QpY constant"""
    bw, bh = (W + 63) // 64 * 8 + 1, (H + 63) // 64 * 8 + 1
    data = np.full((bh, bw), QP << 1, np.int8)
    bs = np.zeros((bh, bw), np.uint8)
    nx, ny = W // N, H // N
    mv = mv.reshape(ny, nx, 2).astype(np.int32)
    cbf = cbf.reshape(ny, nx) != 0
    for by in range(ny):
        for bx in range(nx):
            for dx, dy, shift in ((1, 0, 0), (0, 1, 4)):      # left edge of (bx, by): vertical edge; top edge: horizontal
                px, py = bx - dx, by - dy
                if px < 0 or py < 0:
                    continue
                s = int(cbf[by, bx] or cbf[py, px] or np.abs(mv[by, bx] - mv[py, px]).max() >= 4)
                for k in range(2):      # a 16-sample edge = two 8x8 regions, each with two 4-sample halves
                    ry, rx = 2 * by + (k if dx else 0), 2 * bx + (k if dy else 0)
                    bs[ry, rx] |= (s | s << 2) << shift
    return data.ravel(), bs.ravel()


def _ctus():
    return [(x, y, min(64, W - x), min(64, H - y)) for y in range(0, H, 64) for x in range(0, W, 64)]


def _sao_choice(stats, bd, i):
    """a stand-in for the encoder's SAO decision (turing/EncSao.h:286-520 is floating point and stays on the host): integer only,
    from the statistics.  CTU i takes type i % 3 -- off; band offset: the four bands from the position the statistics name, offsets =
    rounded mean differences clipped to +-7; edge offset: the class whose categories 1..4 carry the largest |sum of differences|,
    offsets = rounded means clipped to 0..7 with the signs the syntax allows"""
    offs = np.zeros(32, np.int16)
    up = bd - min(bd, 10)
    if i % 3 == 0:
        return 0, 0, offs
    if i % 3 == 1:
        start = int(stats[104])
        for k in range(4):
            n, e = int(stats[72 + start + k]), int(stats[40 + start + k])
            offs[(start + k) & 31] = (int(np.sign(e)) * min(7, (abs(e) + n // 2) // n) if n else 0) << up
        return 1, 0, offs
    cls = int(np.argmax([int(np.abs(stats[10 * c + 1:10 * c + 5]).sum()) for c in range(4)]))
    for cat in range(1, 5):
        n, e = int(stats[10 * cls + 5 + cat]), int(stats[10 * cls + cat])
        m = min(7, (abs(e) + n // 2) // n) if n else 0
        offs[cat] = (m if cat <= 2 else -m) << up
    return 2, cls, offs


def _host_picture(R, ref_client, par, src, ref, stride, pus, rq, states, BD):
    """one picture through the reference's functions; returns decisions and every intermediate"""
    res = ref_client.uni(par, src, ref, stride, PAD, pus)
    n = len(pus)
    pred = np.zeros(W * H, src.dtype)
    coef = np.zeros(n * N * N, np.int16)
    level = np.zeros_like(coef)
    deq = np.zeros_like(coef)
    cbf = np.zeros(n, np.int32)
    ssd = np.zeros(n, np.uint32)
    recon = np.zeros_like(src)
    for i, p in enumerate(pus):
        x0, y0 = int(p["x0"]), int(p["y0"])
        qx, qy = int(res["mv"][i][0]), int(res["mv"][i][1])
        so = (y0 + PAD) * stride + x0 + PAD
        R.pred_uni(pred, y0 * W + x0, W, ref, (y0 + (qy >> 2) + PAD) * stride + x0 + (qx >> 2) + PAD, stride, N, N, qx & 3, qy & 3, BD, 8)
        rows = np.arange(N)[:, None]
        r16 = (src[so + rows * stride + np.arange(N)].astype(np.int16) - pred[y0 * W + x0 + rows * W + np.arange(N)].astype(np.int16))
        r16 = np.ascontiguousarray(r16).ravel()
        R.transform(coef, i * N * N, r16, 0, N, 4, 0, BD)
        lv, c = R.rdoq(np.ascontiguousarray(coef[i * N * N:(i + 1) * N * N]), 4, 0, 0, 0, 1, rq["qs"], rq["qshift"], rq["inv"], BD, rq["lam"], states)
        level[i * N * N:(i + 1) * N * N] = lv
        cbf[i] = c
        R.quantize_inverse(deq, i * N * N, level, i * N * N, rq["inv"], rq["dshift"], N * N)
        R.inverse_transform_add(recon, so, stride, pred, y0 * W + x0, W, deq, i * N * N, 4, 0, BD)
        ssd[i] = R.ssd(src, so, stride, recon, so, stride, N, N)
    before = recon.copy()
    data, bs = _strengths(res["mv"], cbf)
    cb = np.full((H // 2) * (W // 2), 128 << (BD - 8), src.dtype)
    cr = cb.copy()
    o = PAD * stride + PAD
    y = recon[o:]          # view whose element 0 is sample (0, 0)
    R.deblock(y, stride, cb, cr, W // 2, W, H, BD, data, bs)
    R.pad_block(recon, o, W, H, stride, PAD, True, True, True, True)      # the edge filter looks one sample beyond a CTU
    # sample-adaptive offset, CTU by CTU: statistics of (source, deblocked), a choice, the filter into a second picture
    sao = np.zeros_like(recon)
    kinds = []
    for i, (x, y0, w, h) in enumerate(_ctus()):
        oc = (y0 + PAD) * stride + x + PAD
        kind, cls, offs = _sao_choice(R.sao_stats(src, oc, stride, recon, oc, stride, w, h, BD), BD, i)
        kinds.append(kind)
        if kind:
            R.sao_filter(sao, oc, stride, recon, oc, stride, w, h, kind, cls, offs if kind == 1 else offs[:5], BD)
        else:
            rows = oc + np.arange(h)[:, None] * stride + np.arange(w)
            sao[rows] = recon[rows]
    R.pad_block(sao, o, W, H, stride, PAD, True, True, True, True)
    return dict(res=res, coef=coef, level=level, cbf=cbf, ssd=ssd, before=before, recon=sao, deblocked=recon, bs=bs, kinds=np.array(kinds))


def _device_picture(hv, L, par, d_src, d_ref, stride, pe, pus, rq, d_states, BD):
    """the same picture on the device; d_src / d_ref: padded planes in HBM (uint8 tensors)"""
    import torch
    from turingcodec_amd import havoc as hmod
    n = len(pus)
    dt = np.uint8 if BD == 8 else np.uint16
    origin = PAD * stride + PAD
    planes = hv.zeros(16 * pe, dt)
    hv.interp_planes_d(BD, planes, pe, d_ref, stride, 12, 4, W + 2 * PAD - 24, H + 2 * PAD - 8)
    with torch.cuda.stream(hv.tstream):
        planes[:d_ref.numel()] = d_ref          # phase 0 = the picture itself
    hv.sync()
    out = np.zeros(n, st.RESULT_DT)
    stats = Stats()
    rc = L.havoc_search_motion_uni(hv.h, np.dtype(dt).itemsize, C.byref(par), C.c_void_p(d_src.data_ptr()), origin, stride, C.c_void_p(d_ref.data_ptr()), origin, stride, PAD,
                                   C.c_void_p(planes.data_ptr()), pe, origin, pus.ctypes.data, n, out.ctypes.data, 8, C.byref(stats))
    assert rc == 0, rc
    mv = out["mv"].astype(np.int32)
    x0, y0 = pus["x0"].astype(np.int64), pus["y0"].astype(np.int64)
    pj = np.zeros((n, 8), np.int32)
    pj[:, 0] = y0 * W + x0
    pj[:, 1] = (y0 + (mv[:, 1] >> 2) + PAD) * stride + x0 + (mv[:, 0] >> 2) + PAD
    pj[:, 2], pj[:, 3], pj[:, 4], pj[:, 5] = N, N, mv[:, 0] & 3, mv[:, 1] & 3
    pred = hv.zeros(W * H, dt)
    hv.pred_uni_d(8, BD, pred, W, d_ref, stride, hv.up(pj), 16, 16)
    fj = np.zeros((n, 4), np.int32)
    fj[:, 0] = np.arange(n) * N * N
    fj[:, 1] = fj[:, 3] = (y0 + PAD) * stride + x0 + PAD
    fj[:, 2] = y0 * W + x0
    d_fj = hv.up(fj)
    coef, level = hv.zeros(n * N * N, np.int16), hv.zeros(n * N * N, np.int16)
    cbf, ssd = hv.zeros(n, np.int32), hv.zeros(n, np.uint32)
    recon = hv.zeros(d_src.numel(), dt)
    hv.tu_forward_d(BD, 0, 4, coef, d_src, stride, pred, W, d_fj)
    jobs = np.zeros(n, hmod.RDOQ_JOB_DT)
    jobs["dst_off"] = jobs["src_off"] = fj[:, 0]
    jobs["quant_scale"], jobs["quant_shift"], jobs["inv_scale"] = rq["qs"], rq["qshift"], rq["inv"]
    jobs["lambda_q16"], jobs["sdh_factor"] = hmod.rdoq_lambda(rq["lam"], rq["inv"])
    jobs["sdh"] = 1
    with torch.cuda.stream(hv.tstream):
        d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(hv.device)
    hv.rdoq_d(BD, 4, level, coef, d_states, d_jobs, cbf, hv.rdoq_workspace(n))
    hv.tu_reconstruct_d(BD, 0, 4, rq["inv"], rq["dshift"], recon, stride, pred, W, d_src, stride, level, d_fj, ssd)
    before = hv.down(recon, dt).copy()
    h_cbf = hv.down(cbf, np.int32)
    data, bs = _strengths(out["mv"], h_cbf)     # the encoder's decisions -> the loop filter's block map (host logic, as in the reference)
    chroma = hv.up(np.full(2 * (H // 2) * (W // 2), 128 << (BD - 8), dt))
    with torch.cuda.stream(hv.tstream):
        d_data = torch.from_numpy(data).to(hv.device)
        d_bs = torch.from_numpy(bs).to(hv.device)
    hv.deblock_d(BD, recon, origin, stride, chroma, 0, (H // 2) * (W // 2), W // 2, W, H, d_data, d_bs)
    hv.pad_block_d(recon, origin, W, H, stride, PAD)
    # sample-adaptive offset: one statistics launch for all CTUs, the choice on the host, one filter launch
    ctus = _ctus()
    sj = np.array([[(y + PAD) * stride + x + PAD] * 2 + [w, h] for x, y, w, h in ctus], np.int32)
    with torch.cuda.stream(hv.tstream):
        d_stats = torch.zeros(105 * len(ctus), dtype=torch.int64, device=hv.device)
    hv._ck(hv.L.havoc_mi355x_sao_stats(hv.h, np.dtype(dt).itemsize, BD, hmod._ptr(d_src), stride, hmod._ptr(recon), stride, hmod._ptr(hv.up(sj)), len(ctus),
                                       hmod._ptr(d_stats)))
    st_all = hv.down(d_stats, np.int64).reshape(-1, 105)
    fj = np.zeros(len(ctus), hmod.SAO_JOB_DT)
    kinds = []
    for i, (x, y, w, h) in enumerate(ctus):
        kind, cls, offs = _sao_choice(st_all[i], BD, i)
        kinds.append(kind)
        fj[i] = (sj[i, 0], sj[i, 0], w, h, kind, cls, offs, (0, 0))
    sao = hv.zeros(d_src.numel(), dt)
    with torch.cuda.stream(hv.tstream):
        d_fj = torch.from_numpy(fj.view(np.uint8).reshape(-1)).to(hv.device)
    hv._ck(hv.L.havoc_mi355x_sao_filter(hv.h, np.dtype(dt).itemsize, BD, hmod._ptr(sao), stride, hmod._ptr(recon), stride, hmod._ptr(d_fj), len(ctus)))
    hv.pad_block_d(sao, origin, W, H, stride, PAD)
    hv.sync()
    return dict(res=out, coef=hv.down(coef, np.int16), level=hv.down(level, np.int16), cbf=h_cbf, ssd=hv.down(ssd, np.uint32), before=before,
                recon=hv.down(sao, dt), deblocked=hv.down(recon, dt), d_recon=sao, bs=bs, stats=stats, kinds=np.array(kinds))


@pytest.mark.parametrize("BD", [8, 10])
def test_two_pictures_through_the_whole_chain_equal_the_reference_functions(BD):
    from turingcodec_amd import havoc as hmod
    from turingcodec_amd.havoc import Havoc
    from turingcodec_amd.workload import dequant_params, picture_lambda, quant_params
    import torch
    planes, stride = st.clip_planes(W, H, 31, BD, PAD)
    src1, ref0, src2 = planes[0], planes[1], planes[2]      # frame 1 predicted from frame 0, then frame 2 from frame 1's reconstruction
    pe = (src1.size + 63) & ~63
    R = reflibs.Reference()
    ref_client = st.Client("ref", 3)
    par = st.medium_params(W, H, BD, QP)
    qs, qshift, _ = quant_params(QP, 4, BD, False)
    inv, dshift = dequant_params(QP, 4, BD)
    rq = dict(qs=qs, qshift=qshift, inv=inv, dshift=dshift, lam=picture_lambda(QP))
    states = R.rdoq_initial_states(QP, 1)

    hv = Havoc(stream="new")
    L = C.CDLL(os.path.join(ROOT, "turingcodec_amd", "libhavoc_search.so"))
    vp, ip, i64 = C.c_void_p, C.c_ssize_t, C.c_int64
    L.havoc_search_motion_uni.argtypes = [vp, C.c_int, C.POINTER(st.Params), vp, i64, ip, vp, i64, ip, C.c_int, vp, ip, i64, vp, C.c_int, vp, C.c_int,
                                          C.POINTER(Stats)]
    with torch.cuda.stream(hv.tstream):
        d_states = torch.from_numpy(states).to(hv.device)

    pus1 = _blocks((-12, -8))      # the clip moves (3, 2) samples per frame
    host1 = _host_picture(R, ref_client, par, src1, ref0, stride, pus1, rq, states, BD)
    dev1 = _device_picture(hv, L, par, hv.up(src1), hv.up(ref0), stride, pe, pus1, rq, d_states, BD)
    for k in ("mv", "mvd", "mv_integer", "mvp_flag", "cost_integer", "cost_subpel", "calls"):
        assert np.array_equal(host1["res"][k], dev1["res"][k]), k
    for k in ("coef", "level", "cbf", "ssd", "bs", "before", "deblocked", "kinds", "recon"):
        assert np.array_equal(host1[k], dev1[k]), k
    assert len(set(host1["kinds"])) == 3 and not np.array_equal(host1["deblocked"], host1["recon"])      # off, band and edge CTUs; samples changed
    # the picture exercised something: vectors off the predictor, coded and uncoded blocks, filtered edges, a filled border
    assert (host1["res"]["mv"] != pus1["mvp"][:, 0]).any() and (host1["cbf"] != 0).any() and (host1["cbf"] == 0).any()
    assert not np.array_equal(host1["before"], host1["recon"]) and host1["bs"].any()
    assert dev1["stats"].launches < 40

    # picture 2 predicts from picture 1's RECONSTRUCTION: on the device straight from the reconstruction plane in HBM
    pus2 = _blocks((-12, -8))
    host2 = _host_picture(R, ref_client, par, src2, host1["recon"], stride, pus2, rq, states, BD)
    dev2 = _device_picture(hv, L, par, hv.up(src2), dev1["d_recon"], stride, pe, pus2, rq, d_states, BD)
    for k in ("mv", "mvd", "cost_integer", "cost_subpel"):
        assert np.array_equal(host2["res"][k], dev2["res"][k]), k
    for k in ("level", "cbf", "ssd", "recon"):
        assert np.array_equal(host2[k], dev2[k]), k
    hv.close()
