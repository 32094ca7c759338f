"""GPU edge cases through the C ABI: empty and ragged batches, argument errors (non-zero return + message, the
reference has no error channel), exact output footprint (no write outside the w x h block: the reference JIT's licence
to over-write to the right, havoc/pred_inter.h:27, is not used), and in-place reconstruction (pred == dst aliasing
allowed by the reference, turing/Reconstruct.cpp:205-207,345)."""
import ctypes as C

import numpy as np
import pytest

import cases
import suite

pytestmark = pytest.mark.gpu
W = cases.PLANE_W


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd import Havoc
    return Havoc(0)


@pytest.fixture(scope="module")
def data():
    return suite.make_inputs(99)


def test_empty_batches_are_noops(hv, data):
    a, b = hv.up(data["u8.a"]), hv.up(data["u8.b"])
    z32, z16 = hv.zeros(64, np.int32), hv.zeros(64, np.int16)
    j = lambda c: hv.up(np.zeros((0, c), np.int32))
    sent = hv.zeros(64, np.int32) + 7
    hv.sad_d(a, W, b, W, j(4), sent)
    hv.sad4_d(a, W, b, W, j(8), sent)
    hv.sad_surface_d(a, W, b, W, 4, 64, 64, j(8), sent)
    hv.ssd_d(a, W, b, W, j(4), sent)
    hv.satd_d(a, W, b, W, j(4), sent)
    hv.satd_multi_d(a, W, b, W, j(20), sent)
    hv.pred_uni_d(8, 8, a, 64, b, W, j(8))
    hv.pred_bi_d(8, 8, a, 64, b, W, j(12))
    hv.subtract_bi_d(8, a, 64, b, W, b, W, j(8))
    hv.subpel_satd_d(8, 8, 64, 64, a, W, b, W, j(8), sent)
    hv.intra_d(8, 3, a, 8, b, j(8))
    hv.intra_satd35_d(8, 3, a, W, b, j(8), sent)
    hv.transform_d(8, 0, 3, z16, z16, 8, j(4))
    hv.inverse_transform_add_d(8, 0, 3, a, 8, a, 8, z16, j(4))
    hv.tu_forward_d(8, 0, 3, z16, a, W, b, W, j(4))
    hv.tu_reconstruct_d(8, 0, 3, 40, 2, a, 8, b, W, a, W, z16, j(4), sent)
    hv.quantize_inverse_d(z16, z16, j(8))
    hv.sync()
    assert (hv.down(sent, np.int32) == 7).all()
    assert np.array_equal(hv.down(a, np.uint8), data["u8.a"])


def test_argument_errors_return_nonzero_with_message(hv, data):
    L, h = hv.L, hv.h
    a = hv.up(data["u8.a"])
    jobs = hv.up(np.zeros((1, 20), np.int32))
    out = hv.zeros(64, np.int32)
    p = lambda t: C.c_void_p(t.data_ptr())
    bad = [
        L.havoc_mi355x_sad(h, 3, p(a), W, p(a), W, p(jobs), 1, p(out)),                       # S must be 1 or 2
        L.havoc_mi355x_sad(h, 1, p(a), W, p(a), W, p(jobs), -1, p(out)),                      # njobs < 0
        L.havoc_mi355x_pred_uni(h, 1, 6, 8, 64, 64, p(a), 64, p(a), W, p(jobs), 1),           # taps
        L.havoc_mi355x_pred_uni(h, 1, 8, 13, 64, 64, p(a), 64, p(a), W, p(jobs), 1),          # bit depth
        L.havoc_mi355x_satd(h, 1, 65, 64, p(a), W, p(a), W, p(jobs), 1, p(out)),              # max_w
        L.havoc_mi355x_sad_surface(h, 1, 97, 64, 64, p(a), W, p(a), W, p(jobs), 1, p(out)),   # range
        L.havoc_mi355x_intra(h, 1, 8, 6, p(a), 8, p(a), p(jobs), 1),                          # log2TrafoSize
        L.havoc_mi355x_transform(h, 8, 1, 3, p(a), p(a), 8, p(jobs), 1),                      # DST exists for 4x4 only
        L.havoc_mi355x_sad(None, 1, p(a), W, p(a), W, p(jobs), 1, p(out)),                    # no context
    ]
    # the round-3 entry points: the device-resident search and the intra decisions
    import ctypes as C2
    from turingcodec_amd.decisions import SearchParams
    i64x2 = (C2.c_int64 * 2)
    def sp(w, h, ctb=64, bd=8):
        return SearchParams(w, h, ctb, 4, 1, 0, 0, 1, 1, bd, 0.1)
    def search(S, par, cx, cy, n=0):
        return L.havoc_mi355x_search_picture_uni(h, S, C2.byref(par), i64x2(1, 1), p(a), 0, W, p(a), i64x2(0, 0), W, p(a), 64, i64x2(0, 0), p(jobs), p(jobs), cx, cy, n,
                                                 p(out), None, p(out), p(out), 0)
    bad += [
        search(3, sp(64, 64), 1, 1),                      # S
        search(1, sp(64, 60), 1, 1),                      # picture height not a multiple of 8
        search(1, sp(128, 64), 1, 1),                     # ctus_x does not match the picture
        search(1, sp(64, 64, ctb=32), 1, 1),              # CTU size
        search(1, sp(64, 64, bd=10), 1, 1),               # bit depth 10 needs S = 2
        search(1, sp(64, 64), 1, 1, n=-1),                # n_pus < 0
        L.havoc_mi355x_intra_expand(h, p(jobs), p(out), p(out), p(out), p(out), 1, 6, 1, 1, 1, 1, 1, 1, p(jobs), p(jobs), p(jobs), p(out), p(out)),   # log2TrafoSize
        L.havoc_mi355x_intra_order(h, p(out), p(jobs), -1, 1, p(out), p(out), p(out), p(out)),                                                     # n < 0
    ]
    assert all(rc != 0 for rc in bad), bad
    assert len(L.havoc_mi355x_last_error()) > 0
    # the context stays usable after errors
    jj = np.array([[cases.off(20, 20), cases.off(30, 30), 16, 16]], np.int32)
    assert hv.sad(data["u8.a"], W, data["u8.b"], W, jj)[0] > 0


@pytest.mark.parametrize("count", [1, 2, 3, 5, 15, 17, 63, 65])
def test_ragged_batches_equal_full_batch_prefix(hv, data, oracle, count):
    """job counts that do not fill a wavefront / workgroup (4, 8, 16 jobs share one): results must not depend on the
    neighbours in the batch"""
    rng = np.random.default_rng(5)
    sizes = [cases.PU_SIZES[i % len(cases.PU_SIZES)] for i in range(70)]
    pairs = np.array([(cases.off(*cases.rand_pos(rng, w, h)), cases.off(*cases.rand_pos(rng, w, h)), w, h) for (w, h) in sizes], np.int32)
    a, b = data["u8.a"], data["u8.x"]
    full = {"sad": hv.sad(a, W, b, W, pairs), "ssd": hv.ssd(a, W, b, W, pairs[pairs[:, 2] == pairs[:, 3]]), "satd": hv.satd(a, W, b, W, pairs)}
    assert np.array_equal(hv.sad(a, W, b, W, pairs[:count]), full["sad"][:count])
    assert np.array_equal(hv.satd(a, W, b, W, pairs[:count]), full["satd"][:count])
    sq = pairs[pairs[:, 2] == pairs[:, 3]][:count]
    assert np.array_equal(hv.ssd(a, W, b, W, sq), full["ssd"][:len(sq)])
    exp = suite.LoopImpl(oracle).sad(a, W, b, W, pairs[:count])
    assert np.array_equal(full["sad"][:count], exp)
    # 35-mode intra search and the fused TU chain with odd counts
    d = {k: v for k, v in data.items()}
    for key in ("u8.intra35", "u8.tuf"):
        got = suite.run(hv, d, keys=[key])
        assert got


def test_prediction_writes_exactly_the_block(hv, data):
    """sentinel-filled destination: only the w x h samples of each job change"""
    rng = np.random.default_rng(3)
    ref = data["u8.a"]
    jobs = []
    for i, (w, h) in enumerate(cases.PU_SIZES + [(6, 8), (2, 4), (12, 16)]):
        x, y = cases.rand_pos(rng, w, h)
        jobs.append((i * suite.SLOT + 64 + 3, cases.off(x, y), w, h, i % 4, (i // 4) % 4, 0, 0))
    jobs = np.array(jobs, np.int32)
    dst = hv.zeros(len(jobs) * suite.SLOT + 8192, np.uint8) + 0xA5
    for idx, mw, mh in hv.size_classes(jobs[:, 2], jobs[:, 3]):
        hv.pred_uni_d(8, 8, dst, 64, hv.up(ref), W, hv.up(np.ascontiguousarray(jobs[idx])), mw, mh)
    out = hv.down(dst, np.uint8)
    touched = np.zeros(len(out), bool)
    for (do, _, w, h, *_rest) in jobs.tolist():
        for r in range(h):
            touched[do + r * 64:do + r * 64 + w] = True
    assert (out[~touched] == 0xA5).all()
    assert (out[touched] != 0xA5).any()


def test_reconstruction_in_place(hv, data, oracle):
    """inverse_transform_add and tu_reconstruct with pred == dst (the intra path reconstructs into the plane it
    predicted into)"""
    rng = np.random.default_rng(8)
    n, log2 = 16, 4
    m = 9
    plane = data["u8.a"].copy()
    coef = rng.integers(-300, 300, m * n * n).astype(np.int16)
    offs = [cases.off(16 + 20 * (i % 3) + 1, 16 + 20 * (i // 3) + 1) for i in range(m)]
    jobs = np.array([(i * n * n, 0, offs[i], offs[i]) for i in range(m)], np.int32)   # tu_job: coef, res, pred, dst
    exp = plane.copy()
    for i in range(m):
        oracle.inverse_transform_add(exp, offs[i], W, exp, offs[i], W, coef, i * n * n, log2, 0, 8)
    d = hv.up(plane)
    hv.inverse_transform_add_d(8, 0, log2, d, W, d, W, hv.up(coef), hv.up(jobs))
    assert np.array_equal(hv.down(d, np.uint8), exp)
    assert not np.array_equal(exp, plane)
