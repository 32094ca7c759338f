"""CPU checks of the bench workload (turingcodec_amd/workload.py): the call counts are the survey's, and every job
stays inside the padded picture store (the kernels do no bounds checking, as the reference's primitives)."""
import numpy as np
import pytest

from turingcodec_amd.workload import CALLS_1080P, FrameWorkload


@pytest.fixture(scope="module")
def wl():
    return FrameWorkload(640, 360, 8, 3)


def test_counts_scale_with_ctu_count(wl):
    f = (10 * 6) / 510.0
    for k, v in CALLS_1080P.items():
        assert abs(wl.counts[k] - v * f) <= 1, k
    assert len(wl.sad4) == wl.counts["sad4"] and len(wl.sad) == wl.counts["sad"]
    assert sum(len(g["jobs"]) for g in wl.tu.values()) == wl.counts["tu"]
    assert sum(len(j) for j in wl.intra_search.values()) * 35 <= wl.counts["intra_satd"] + 35
    nsub = sum(len(j) for j in wl.subpel.values())
    assert nsub == sum(int(j[:, 3].sum()) for j in wl.subpel_planes.values())   # same candidates through both routes
    assert nsub + len(wl.uni8) == sum(wl.counts[k] for k in ("uni8_hv", "uni8_h", "uni8_v", "uni8_copy"))


def _inside(wl, off, w, h, reach, planes, plane_len, stride, rows):
    off = np.asarray(off, np.int64)
    p = off // plane_len
    r = off % plane_len
    y, x = r // stride, r % stride
    assert (p >= 0).all() and (p < planes).all()
    assert (x - reach >= 0).all() and (x + w + reach + 3 <= stride).all()
    assert (y - reach >= 0).all() and (y + h + reach <= rows).all()


def test_luma_jobs_stay_inside_the_padded_planes(wl):
    pl, st = wl.plane_len, wl.stride
    rows = pl // st
    j = wl.sad4
    for k in range(1, 5):
        _inside(wl, j[:, k], j[:, 5], j[:, 6], 0, 3, pl, st, rows)
    _inside(wl, j[:, 0], j[:, 5], j[:, 6], 0, 3, pl, st, rows)
    _inside(wl, wl.uni8[:, 1], wl.uni8[:, 2], wl.uni8[:, 3], 4, 3, pl, st, rows)          # 8-tap reach -3 .. +4
    for c in (1, 2):
        _inside(wl, wl.bi8[:, c], wl.bi8[:, 3], wl.bi8[:, 4], 4, 3, pl, st, rows)
    for j in wl.subpel.values():
        _inside(wl, j[:, 1], j[:, 2], j[:, 3], 4, 3, pl, st, rows)
    m = wl.me_search
    _inside(wl, m[:, 1], m[:, 2], m[:, 3], 64, 3, pl, st, rows)                           # surfaces up to +-64
    # candidates against the phase planes: inside the rectangle interp_planes fills (picture + plane_margin)
    lo, hi_x, hi_y = 96 - wl.plane_margin, 96 + wl.width + wl.plane_margin, 96 + wl.height + wl.plane_margin
    for j in wl.subpel_planes.values():
        for k in range(16):
            use = j[:, 3] > k
            r = j[use, 4 + k].astype(np.int64) % pl
            y, x = r // st, r % st
            assert (x >= lo).all() and (y >= lo).all()
            assert (x + j[use, 1] <= hi_x).all() and (y + j[use, 2] <= hi_y).all()


def test_jobs_are_grouped_as_the_search_issues_them(wl):
    j = wl.sad4
    run = 31
    k = (len(j) // run) * run
    g = j[:k].reshape(-1, run, 8)
    assert (g[:, :, 0] == g[:, :1, 0]).all() and (g[:, :, 5] == g[:, :1, 5]).all()       # one PU per run of SAD4 calls
    for jm in wl.subpel_planes.values():
        assert ((jm[:, 3] == 16) | (jm[:, 3] == 1)).all()


@pytest.mark.parametrize("res", [(640, 360), (1920, 1080)])
def test_final_reconstruction_pass_tiles_the_picture_once(res):
    """workload.recon: every luma / chroma sample of the picture belongs to exactly one transform unit of the final pass"""
    w, h = res
    wl = FrameWorkload(w, h, 8, 5)
    for comp, (pw, ph, stride, plane_len, pad, plane) in {"y": (w, h, wl.stride, wl.plane_len, 96, 3), "cb": (w // 2, h // 2, wl.cstride, wl.cplane_len, 48, 3),
                                                          "cr": (w // 2, h // 2, wl.cstride, wl.cplane_len, 48, 4)}.items():
        cover = np.zeros((ph, pw), np.int32)
        for log2, g in wl.recon[comp].items():
            n = g["n"]
            assert n == 1 << log2 and len(g["levels"]) == len(g["jobs"]) * n * n
            o = g["jobs"][:, 3].astype(np.int64) - plane * plane_len
            y, x = o // stride - pad, o % stride - pad
            for xi, yi in zip(x, y):
                cover[yi:yi + n, xi:xi + n] += 1
            _inside(wl, g["jobs"][:, 2], np.full(len(o), n), np.full(len(o), n), 0, 5, plane_len, stride, plane_len // stride)
        assert cover.min() == 1 and cover.max() == 1, comp


def test_all_intra_mix_has_no_inter_calls_and_the_survey_counts():
    from turingcodec_amd.workload import CALLS_AI_PER_CTU
    wl = FrameWorkload(640, 360, 8, 7, mix="ai")
    assert set(wl.counts) == set(CALLS_AI_PER_CTU)
    for k, v in CALLS_AI_PER_CTU.items():
        assert abs(wl.counts[k] - v * 60) <= 1, k
    assert sum(len(g["jobs"]) for g in wl.tu.values()) == wl.counts["tu"]


def test_quantiser_parameters_follow_qpstate():
    """turing/QpState.h:85-94 / Reconstruct.cpp:286,311,315 at the QPs of BASELINE.json's configs"""
    from turingcodec_amd.workload import dequant_params, quant_params
    assert quant_params(32, 3, 8, False) == (20560, 29 - 8 + 5 - 3, 85 << 7)
    # 10-bit: QP' = QP + QpBdOffsetY = 27 + 12 = 39 (turing/QpState.h:56, 79-94)
    assert quant_params(27, 5, 10, True) == (18396, 29 - 10 + 6 - 5, 171 << 7)
    assert dequant_params(32, 3, 8) == (51 << 5, 2) and dequant_params(27, 2, 10) == (57 << 6, 3)


def test_yuv_reader_frames_and_planes(tmp_path):
    """raw planar 4:2:0 input (turing/encode.cpp:600-640): frame count, plane views, partial last frame, size errors"""
    import pytest
    from turingcodec_amd.picture_io import YuvReader
    from turingcodec_amd.workload import synth_frames
    W, H = 64, 48
    for bd in (8, 10):
        frames = synth_frames(W, H, 2, 3, bd)
        path = tmp_path / f"c{bd}.yuv"
        with open(path, "wb") as f:
            for fr in frames:
                for plane in fr:
                    f.write(np.ascontiguousarray(plane, np.uint8 if bd == 8 else "<u2").tobytes())
            f.write(b"\x01" * 17)
        rd = YuvReader(str(path), W, H, bd)
        assert len(rd) == 2 and rd.frame_bytes == W * H * 3 // 2 * (1 if bd == 8 else 2)
        for i in range(2):
            assert len(rd[i]) == rd.frame_bytes
            for c in range(3):
                assert np.array_equal(rd.planes(i)[c], frames[i][c])
        with pytest.raises(IndexError):
            rd[2]
    with pytest.raises(ValueError):
        YuvReader(str(path), 63, 48, 10)
    with pytest.raises(ValueError):
        YuvReader(str(path), 1920, 1080, 10)
