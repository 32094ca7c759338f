"""havoc_mi355x_intra_measure -- the 35-mode stage of ONE intra partition in one launch (what libhavoc_classic.so's serve layer precomputes in the wait of a partition's
first call: turing/Search.hpp:113-142, Reconstruct.cpp:244-353) -- against the entry points whose values it claims to give, each of which is held against the oracle and
the reference's goldens elsewhere: havoc_mi355x_satd on the partition's tiles, havoc_mi355x_tu_forward (DST-VII and DCT for 4x4), havoc_mi355x_tu_reconstruct on zero
levels.  Bit exact, 8- and 10-bit, every size, a ragged last wavefront."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
W = cases.PLANE_W


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd import Havoc
    return Havoc(0)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
@pytest.mark.parametrize("with_satd", [True, False])
def test_intra_measure_equals_the_separate_entry_points(hv, bd, log2, with_satd):
    rng = np.random.default_rng(100 * bd + log2)
    S = 1 if bd == 8 else 2
    n, modes = 1 << log2, 35
    ts = 8 if n >= 8 else 4
    tx = n // ts
    tiles = tx * tx
    src = cases.rand_plane(rng, S, bd)
    pred_stride = n * modes + 4
    pred = rng.integers(0, 1 << bd, (n, pred_stride)).astype(src.dtype)       # one row of 35 prediction blocks side by side, as the serve layer lays them out
    pred[:, :n] = (1 << bd) - 1                                                # a saturated and an empty candidate
    pred[:, n:2 * n] = 0
    sx, sy = 24, 40
    src_off = cases.off(sx, sy)
    jobs = np.array([(i * n * n, src_off, i * n, i * n * n) for i in range(modes)], np.int32)     # tu_fused_job: coef, src, pred, rec
    d_src, d_pred, d_jobs = hv.up(src), hv.up(pred), hv.up(jobs)
    co, co_dct = hv.zeros(modes * n * n, np.int16), hv.zeros(modes * n * n, np.int16)
    satd = hv.zeros(modes * tiles, np.int32)
    rec0 = hv.zeros(modes * n * n, src.dtype)
    ssd0 = hv.zeros(modes, np.uint32)
    hv.intra_measure_d(bd, log2, co, co_dct if log2 == 2 else None, satd if with_satd else None, rec0, ssd0, d_src, W, d_pred, pred_stride, d_jobs, with_satd)

    # forward transforms
    want = hv.zeros(modes * n * n, np.int16)
    hv.tu_forward_d(bd, 1 if log2 == 2 else 0, log2, want, d_src, W, d_pred, pred_stride, d_jobs)
    assert np.array_equal(hv.down(co, np.int16), hv.down(want, np.int16))
    if log2 == 2:
        want = hv.zeros(modes * n * n, np.int16)
        hv.tu_forward_d(bd, 0, log2, want, d_src, W, d_pred, pred_stride, d_jobs)
        assert np.array_equal(hv.down(co_dct, np.int16), hv.down(want, np.int16))
    # tile SATDs (source tile against prediction tile; `satd` takes both operands with one stride each)
    if with_satd:
        pair = np.array([(src_off + (t // tx) * ts * W + (t % tx) * ts, i * n + (t // tx) * ts * pred_stride + (t % tx) * ts, ts, ts) for i in range(modes) for t in range(tiles)],
                        np.int32)
        out = hv.zeros(len(pair), np.int32)
        hv.satd_d(d_src, W, d_pred, pred_stride, hv.up(pair), out, 8, 8)
        assert np.array_equal(hv.down(satd, np.int32), hv.down(out, np.int32))
    # reconstruction from zero levels + its SSD
    rec = hv.zeros(modes * n * n, src.dtype)
    ssd = hv.zeros(modes, np.uint32)
    hv.tu_reconstruct_d(bd, 1 if log2 == 2 else 0, log2, 40, 3, rec, n, d_pred, pred_stride, d_src, W, hv.zeros(modes * n * n, np.int16), d_jobs, ssd)
    assert np.array_equal(hv.down(rec0, src.dtype), hv.down(rec, src.dtype))
    assert np.array_equal(hv.down(ssd0, np.uint32), hv.down(ssd, np.uint32))
    assert hv.down(ssd0, np.uint32).max() > 0
