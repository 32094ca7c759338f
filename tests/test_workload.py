"""CPU checks of the bench workload (turingcodec_amd/workload.py): the call counts and size mixes are the MEASURED ones of the reference's own
encoder (profiles/r04_reference_call_mix_1080p.json, written by profiles/measure_call_mix.py), and every job stays inside the padded picture
store (the kernels do no bounds checking, as the reference's primitives)."""
import json
import os

import numpy as np
import pytest

from turingcodec_amd import workload
from turingcodec_amd.workload import CALLS_1080P, FrameWorkload

MIX = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_reference_call_mix_1080p.json")


def test_constants_are_the_committed_measurement():
    """`config.workload` says "measured mix" -- this is where that is checked"""
    m = json.load(open(MIX))
    c, by, s = m["calls_by_entry_point"], m["by_size"], m["searches"]
    assert m["case"] == "mix_1080p_qp32" and m["clip"].startswith("1920x1080, 9 frames")
    per_b = {"sad4": c["sad4"], "sad": c["sad"], "uni8_hv": c["uni8_hv"], "uni8_h": c["uni8_h"], "uni8_v": c["uni8_v"], "uni8_copy": c["uni8_copy"],
             "uni4_h": c["uni4_h"], "uni4_v": c["uni4_v"], "uni4_hv": c["uni4_hv"], "uni4_copy": c["uni4_copy"], "bi8": c["pred_bi8"], "bi4": c["pred_bi4"],
             "subtract_bi": c["subtract_bi"], "searches": s["searchMotionUni"], "subpel": s["costDistortionMv calls (interpolate + SATD)"]}
    for k, v in per_b.items():
        assert CALLS_1080P[k] == v // 8, k
    parts = {int(k): v for k, v in s["intra_partitions_by_log2_size"].items()}
    blocks = {5: parts[5] + 4 * parts[6], 4: parts[4], 3: parts[3], 2: parts[2]}          # a 64x64 partition is predicted as four 32x32 blocks
    assert CALLS_1080P["intra_satd"] == 35 * sum(blocks.values()) // 9 and CALLS_1080P["intra_rd"] == (c["intra"] - 35 * sum(blocks.values())) // 9
    assert CALLS_1080P["tu"] == c["transform"] // 9 == c["inverse_transform"] // 9 and CALLS_1080P["ssd"] == c["ssd"] // 9
    assert s["searchMotionBi"] == c["subtract_bi"] and s["predictIntraLuma calls of the 35-mode stage"] == 35 * sum(parts.values())
    # sizes: only square units are searched; the shares are the measured ones
    assert set(s["uni_searches_by_size"]) == {"8x8", "16x16", "32x32", "64x64"}
    assert {(w, h): n for w, h, n in workload.PU_MIX} == {tuple(int(v) for v in k.split("x")): n for k, n in s["uni_searches_by_size"].items()}
    assert dict(workload.INTRA_MIX) == blocks
    assert dict(workload.INTRA_RD_MIX) == {l: by["intra"][f"{1 << l}x{1 << l}"] - 35 * blocks[l] for l in blocks}
    t = by["transform"]
    assert {(l, tr): n for l, tr, n in workload.TU_MIX} == {(5, 0): t["32x32"], (4, 0): t["16x16"], (3, 0): t["8x8"], (2, 0): t["4x4"] - c["transform_dst"],
                                                           (2, 1): c["transform_dst"]}
    uni_sad4 = c["sad4"] - 33 * s["searchMotionBi"]                                        # a bi-directional refinement's grid is 11 rows x 3 calls
    assert abs(workload.SAD4_PER_SEARCH - uni_sad4 / s["searchMotionUni"]) < 1
    # 4 x 4 is 65 % of the intra predictions, 53 % of the transforms (the round-3 workload assumed 5 % / the survey's smoother clip)
    assert 0.6 < by["intra"]["4x4"] / c["intra"] < 0.7


def test_picture_units_follow_the_measured_sizes():
    """workload.picture_pus (the decision-driven path's units): 2Nx2N only, counts per size within 20 % of the measured searches per B picture and list"""
    s = json.load(open(MIX))["searches"]["uni_searches_by_size"]
    pus, first, cx, cy = workload.picture_pus(1920, 1080, 11)
    assert (pus["part_2Nx2N"] == 1).all() and (pus["w"] == pus["h"]).all()
    for size in (16, 32, 64):
        want = s[f"{size}x{size}"] / 16                                                    # 8 B pictures x 2 lists
        assert abs(int((pus["w"] == size).sum()) - want) < 0.2 * want, size
    assert abs(len(pus) - sum(s.values()) / 16) < 0.2 * sum(s.values()) / 16


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
    run = workload.SAD4_PER_SEARCH
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
