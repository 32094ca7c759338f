"""The PRODUCER's half of CTU-row bands (DecisionPicture.step_banded; turing/TaskDeblock.cpp:151-167 -- a picture's rows are deblocked, padded and published while the rows
below are still being encoded): the search kernel runs on the picture's stream and on a second stream every band of CTU rows is queued behind a launch that ends when
the band's rows (and the row below them) are searched -- merge candidates, prediction, transform trees, chroma, block structure, boundary strengths, the band's
deblocking and padding, the counter of final rows.  Everything the step leaves must equal what step() leaves: search results, motion field, bi-directional refinements,
merge decisions, the block structure, and the padded luma and chroma reconstructions sample for sample."""
import time

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("res,bit_depth,rows", [((1920, 1080), 8, 4), ((832, 480), 10, 2), ((1280, 720), 8, 1)])
def test_the_banded_step_leaves_what_the_whole_step_leaves(res, bit_depth, rows):
    import torch
    from turingcodec_amd import Havoc
    from turingcodec_amd.decisions import DecisionPicture
    W, H = res
    hv = Havoc(0, stream=torch.cuda.Stream(device=0).cuda_stream)
    side_stream = torch.cuda.Stream(device=0, priority=-1)      # another priority = its own hardware queue (tests/test_search_gate.py)
    side = Havoc(0, stream=side_stream.cuda_stream)
    whole = DecisionPicture(hv, W, H, bit_depth, 32, seed=11)
    want, want_field, _ = whole.step()
    whole.step()      # (the second call records the launches after the searches into a graph; the third one is the step as the bench times it)
    hv.sync()
    t = time.perf_counter()
    whole.step()
    hv.sync()
    t_whole = time.perf_counter() - t
    banded = DecisionPicture(hv, W, H, bit_depth, 32, seed=11)
    rows_final = hv.zeros(1, np.int32)
    banded.step_banded(side, rows, rows_final)
    banded.step_banded(side, rows, rows_final)      # (the second call records each band's launches into a graph; from the third on they are replayed)
    with torch.cuda.stream(hv.tstream):
        banded.recon.zero_()
        banded.crecon.zero_()
    t = time.perf_counter()
    got, got_field, _ = banded.step_banded(side, rows, rows_final)
    t_banded = time.perf_counter() - t
    print(f"{W}x{H} {bit_depth}-bit: whole step {1e3 * t_whole:.2f} ms, banded ({rows} CTU rows) {1e3 * t_banded:.2f} ms")
    assert got.tobytes() == want.tobytes() and np.array_equal(got_field, want_field)
    assert banded.bi_results.tobytes() == whole.bi_results.tobytes()
    assert int(hv.down(rows_final, np.int32)[0]) == H + banded.PAD
    for name in ("recon", "crecon", "pred", "cpred", "d_cells", "d_data", "d_bs"):
        with torch.cuda.stream(hv.tstream):
            a, b = getattr(whole, name).cpu().numpy(), getattr(banded, name).cpu().numpy()
        assert np.array_equal(a, b), (name, np.flatnonzero(a != b)[:8], len(np.flatnonzero(a != b)))
    # the merge decisions of the units, band by band against the whole picture's
    m = whole.merge
    at = {(int(u["x0"]), int(u["y0"])): i for i, u in enumerate(whole.units)}
    for v in banded._views:
        mv = v.merge
        idx = np.array([at[(int(u["x0"]), int(u["y0"]))] for u in v.units])
        for k in ("vectors", "satd", "cost", "best"):
            assert np.array_equal(mv[k], m[k][idx]), (k, v.band_span)


@pytest.mark.gpu
def test_a_picture_is_searched_while_the_picture_it_predicts_from_is_still_being_decided():
    """Both halves together, the anchor chain of one sequence in small: picture B predicts (list 0) from the RECONSTRUCTION of picture A.  A runs step_banded; as each
    of its bands becomes final, A's side stream hands the rows on to B -- into B's reference plane, then the fractional planes of the rows whose filter taps are there,
    then B's gate -- and B's search, launched at the same time as A's, follows A down the picture.  B must find what it finds when it starts after A is complete, and
    the pair must take less time than one after the other."""
    import threading
    import torch
    from turingcodec_amd import Havoc
    from turingcodec_amd.decisions import DecisionPicture
    W, H, bands = 1920, 1080, 2
    hvA = Havoc(0, stream=torch.cuda.Stream(device=0).cuda_stream)
    hvB = Havoc(0, stream=torch.cuda.Stream(device=0).cuda_stream)
    side_stream = torch.cuda.Stream(device=0, priority=-1)
    side = Havoc(0, stream=side_stream.cuda_stream)
    A = DecisionPicture(hvA, W, H, 8, 32, seed=11, intra=False)
    B = DecisionPicture(hvB, W, H, 8, 32, seed=12, intra=False)
    pe, PAD, stride = B.pe, B.PAD, B.stride
    rows_total = H + 2 * PAD
    ref0 = B.d_pic[pe:2 * pe]
    ph0 = B.d_phase[:16 * pe]

    def b_after_a():
        with torch.cuda.stream(hvB.tstream):
            ref0.copy_(A.recon[:pe])
        B.phase_planes()
        return B.search()

    # ---- one after the other
    A.step_banded(side, bands)
    hvA.sync()
    want, want_field, _ = b_after_a()
    want_bi = B.bi_results.copy()
    hvB.sync()
    t = time.perf_counter()
    A.step_banded(side, bands)
    t_a = time.perf_counter() - t
    b_after_a()
    hvB.sync()
    serial = time.perf_counter() - t

    # ---- together
    def run_pair():
        gate = hvB.zeros(2, np.int32)
        with torch.cuda.stream(hvB.tstream):
            ref0.zero_()
            ph0.zero_()
            gate[1:].fill_(1 << 20)      # list 1's picture is complete (its planes were made above)
        hvB.sync()
        hvB.search_gate(gate)
        state = {"rows": 0, "planes": 4}

        def hand_on(b, final):
            upto = rows_total if final >= H + PAD else PAD + final      # padded rows of A's reconstruction that are final
            lo, end = state["rows"], (rows_total - 4 if upto == rows_total else upto - 4)
            with torch.cuda.stream(side.tstream):
                ref0[lo * stride:upto * stride] = A.recon[lo * stride:upto * stride]
                ph0[lo * stride:upto * stride] = A.recon[lo * stride:upto * stride]
            side.interp_planes_d(B.bd, ph0, pe, ref0, stride, 12, state["planes"], W + 2 * PAD - 24, end - state["planes"])
            with torch.cuda.stream(side.tstream):
                gate[:1].fill_(end - PAD)
            state["rows"], state["planes"] = upto, end

        out = {}

        def search_b():
            try:
                out["B"] = B.search()
                out["bi"] = B.bi_results.copy()
            except BaseException as e:
                out["error"] = e

        th = threading.Thread(target=search_b)
        t0 = time.perf_counter()
        th.start()
        out["A"] = A.step_banded(side, bands, on_band=hand_on)
        th.join()
        hvB.sync()
        out["seconds"] = time.perf_counter() - t0
        hvB.search_gate(None)
        return out

    run_pair()
    out = run_pair()
    assert "error" not in out, repr(out.get("error"))
    got, got_field, _ = out["B"]
    assert got.tobytes() == want.tobytes() and np.array_equal(got_field, want_field) and out["bi"].tobytes() == want_bi.tobytes()
    print(f"A {1e3 * t_a:.2f} ms, then B: {1e3 * serial:.2f} ms together; B following A down the picture: {1e3 * out['seconds']:.2f} ms")
    # A's first CTU row is half of A's time (a row is 30 CTUs one after the other, the rows follow two CTUs apart): what B can hide of A is what comes after A's first band
    assert out["seconds"] < serial - 0.05 * t_a, (out["seconds"], serial, t_a)      # (measured: 48.2 ms against 53.2 with A = 15.7)
