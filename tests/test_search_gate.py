"""Searching into reference pictures that are STILL ARRIVING (havoc_mi355x_search_gate; VERDICT r4 missing #4 / next #6, the consumer's half): the reference's rule is that a
CTU starts once its reference picture is reconstructed three CTU rows below it (turing/TaskEncodeSubstream.cpp:71-95; TaskDeblock.cpp:151-167 publishes the deblocked,
padded rows), which across GPUs means: a picture's searches may run while the later BANDS of its references are still on their way (frame_parallel.BandPlan /
ReferenceExchange.send_band).  Here both reference pictures of a 1080p decision picture are wiped, the search kernel is launched, and a second stream then delivers
them band by band -- the picture rows, then the 15 fractional planes of the rows that have their filter taps, then the counter the kernel polls -- with pauses between
the bands.  Every search result, the motion field and every bi-directional refinement must be what the search finds in complete references, and most of the search
must have run BEFORE the last band arrived."""
import threading
import time

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("res,bit_depth", [((1920, 1080), 8), ((832, 480), 10), ((832, 480), 8), ((1280, 720), 10)])
def test_searches_run_while_the_bands_of_their_references_arrive(res, bit_depth):
    import torch
    from turingcodec_amd import Havoc
    from turingcodec_amd.decisions import DecisionPicture
    from turingcodec_amd.frame_parallel import BandPlan
    W, H = res
    s = torch.cuda.Stream(device=0)
    hv = Havoc(0, stream=s.cuda_stream)
    dp = DecisionPicture(hv, W, H, bit_depth, 32, seed=11, intra=False)
    assert dp.params.concurrent_frames > 1
    pe, PAD, stride = dp.pe, dp.PAD, dp.stride
    dp.phase_planes()
    want, want_field, _ = dp.search()
    want_bi = dp.bi_results.copy()
    hv.sync()
    t = time.perf_counter()
    dp.search()
    hv.sync()
    alone = time.perf_counter() - t

    # ---- the references leave ...
    saved = [dp.d_pic[(1 + r) * pe:(2 + r) * pe].clone() for r in (0, 1)]
    with torch.cuda.stream(hv.tstream):
        dp.d_pic[pe:3 * pe].zero_()
        dp.d_phase.zero_()
    gate = hv.zeros(2, np.int32)
    hv.sync()
    hv.search_gate(gate)

    # ---- ... and come back band by band on another stream
    # (a stream of ANOTHER PRIORITY: HIP multiplexes the streams of one priority onto a few hardware queues, and a kernel that waits blocks whatever is queued behind it on
    # its queue -- with both streams at one priority this test's delivery sat behind the waiting search, which gave up after its 8 seconds: gpu call r05s4)
    prod_stream = torch.cuda.Stream(device=0, priority=-1)
    prod = Havoc(0, stream=prod_stream.cuda_stream)
    plan = BandPlan(H, PAD, stride, stride // 2, band_ctu_rows=3 if H > 600 else 2)
    rows_total = H + 2 * PAD
    pause = max(0.004, 1.2 * alone / plan.n_bands)
    marks = {}

    def deliver():
        try:
            deliver_bands()
        except BaseException as e:      # (a delivery that dies must not leave the search waiting for its time-out: let it through, the test fails on the exception)
            marks["error"] = e
            with torch.cuda.stream(prod.tstream):
                gate.fill_(1 << 20)
            prod.sync()

    def deliver_bands():
        time.sleep(0.003)      # the search kernel is in flight first
        done = 4
        for b in range(plan.n_bands):
            lo, hi = plan.luma_rows(b)
            end = rows_total - 4 if b == plan.n_bands - 1 else hi - 4      # rows whose 8-tap filters have all their rows
            with torch.cuda.stream(prod.tstream):
                for r in (0, 1):
                    ref = dp.d_pic[(1 + r) * pe:(2 + r) * pe]
                    ph = dp.d_phase[r * 16 * pe:(r + 1) * 16 * pe]
                    ref[lo * stride:hi * stride] = saved[r][lo * stride:hi * stride]
                    ph[lo * stride:hi * stride] = saved[r][lo * stride:hi * stride]          # phase 0 = the picture itself
            for r in (0, 1):
                ref = dp.d_pic[(1 + r) * pe:(2 + r) * pe]
                ph = dp.d_phase[r * 16 * pe:(r + 1) * 16 * pe]
                prod.interp_planes_d(dp.bd, ph, pe, ref, stride, 12, done, W + 2 * PAD - 24, end - done)
            with torch.cuda.stream(prod.tstream):
                gate.fill_(end - PAD)      # picture rows [.., end - PAD) are final in the picture and in every plane
            prod.sync()
            marks[b] = time.perf_counter()
            done = end
            if b < plan.n_bands - 1:
                time.sleep(pause)

    th = threading.Thread(target=deliver)
    t0 = time.perf_counter()
    th.start()
    try:
        got, got_field, _ = dp.search()
    except RuntimeError as e:
        th.join()
        raise AssertionError((str(e), "gate", hv.down(gate, np.int32).tolist(), "bands delivered at", {k: round(v - t0, 4) if isinstance(v, float) else repr(v) for k, v in marks.items()},
                              "failed after", round(time.perf_counter() - t0, 3), "bands", [plan.luma_rows(b) for b in range(plan.n_bands)], "alone", alone))
    got_bi = dp.bi_results.copy()
    hv.sync()
    t_done = time.perf_counter()
    th.join()
    hv.search_gate(None)
    assert "error" not in marks, repr(marks["error"])

    assert got.tobytes() == want.tobytes(), np.flatnonzero(got["mv"] != want["mv"])[:8]
    assert np.array_equal(got_field, want_field)
    assert got_bi.tobytes() == want_bi.tobytes()
    t_last = marks[plan.n_bands - 1]
    # the delivery took longer than a search alone, so a search that had WAITED for complete references would end `alone` after the last band; this one had most
    # of its rows behind it by then
    assert t_last - t0 > alone, (t_last - t0, alone)
    if H > 600:      # (17 CTU rows: the four that reach into the last band are a wavefront of 36 CTU steps against the picture's 62; a small picture is mostly its last band)
        assert t_done - t_last < 0.95 * alone, (t_done - t_last, alone, pause, plan.n_bands)      # (measured: 0.6-0.7)
    # and with the references back the ungated search is unchanged
    again, again_field, _ = dp.search()
    assert again.tobytes() == want.tobytes() and np.array_equal(again_field, want_field)


@pytest.mark.gpu
def test_the_gate_is_refused_where_vectors_are_not_limited():
    import torch
    from turingcodec_amd import Havoc
    from turingcodec_amd.decisions import DecisionPicture
    hv = Havoc(0, stream=torch.cuda.Stream(device=0).cuda_stream)
    dp = DecisionPicture(hv, 416, 240, 8, 32, seed=3, intra=False)
    dp.phase_planes()
    dp.params.concurrent_frames = 1      # one frame at a time: the search does not limit its vectors to what a gate could promise
    gate = hv.zeros(2, np.int32)
    hv.search_gate(gate)
    with pytest.raises(RuntimeError):
        dp.search()
    hv.search_gate(None)
    dp.search()
