"""Deblocking: the CPU oracle against the reference's OWN LoopFilter templates (compiled from its header by
oracle/Makefile, driven in the reference's CTU order), and the GPU kernel against the oracle."""
import numpy as np
import pytest

import cases


def block_map(rng, width, height, qp_lo=20, qp_hi=45, p_edge=0.6, p_disabled=0.04):
    """LoopFilter::Block arrays on the ((W+63)/64*8 + 1) x ((H+63)/64*8 + 1) grid: QpY and the filter-disabled bit per 8x8 region,
    2-bit strengths for its left edge (rows 0-3, 4-7) and top edge (columns 0-3, 4-7); no strength on the picture boundary"""
    bw, bh = (width + 63) // 64 * 8 + 1, (height + 63) // 64 * 8 + 1
    data = ((rng.integers(qp_lo, qp_hi + 1, (bh, bw)) << 1) | (rng.random((bh, bw)) < p_disabled)).astype(np.int8)
    bs = np.zeros((bh, bw), np.uint8)
    for k in range(4):
        v = np.where(rng.random((bh, bw)) < p_edge, rng.integers(1, 3, (bh, bw)), 0)
        bs |= (v << (2 * k)).astype(np.uint8)
    bs[:, 0] &= 0xF0    # left picture edge: no vertical-edge strength
    bs[0, :] &= 0x0F    # top picture edge
    return data.ravel(), bs.ravel()


def planes(rng, width, height, S, bd, kind):
    mx = (1 << bd) - 1
    dt = cases.sample_dtype(S)

    def one(w, h):
        if kind == "smooth":   # blocky but smooth content: what the filter is for (strong and normal filters both trigger)
            base = np.kron(rng.integers(0, mx + 1, (h // 8 + 1, w // 8 + 1)), np.ones((8, 8)))[:h, :w]
            return np.clip(base * 0.2 + mx * 0.4 + rng.integers(-2, 3, (h, w)), 0, mx).astype(dt)
        if kind == "extremes":
            return (rng.integers(0, 2, (h, w)) * mx).astype(dt)
        return rng.integers(0, mx + 1, (h, w)).astype(dt)
    return one(width, height), one(width // 2, height // 2), one(width // 2, height // 2)


CASES = [(64, 64, 1, 8), (208, 120, 1, 8), (136, 72, 2, 10), (320, 192, 2, 9), (8, 8, 1, 8)]


@pytest.mark.parametrize("width,height,S,bd", CASES)
@pytest.mark.parametrize("kind", ["smooth", "uniform", "extremes"])
def test_oracle_deblock_equals_the_reference_templates(oracle, reference_c, width, height, S, bd, kind):
    rng = np.random.default_rng(width * 7 + height + bd)
    for tc2, beta2, cbq, crq in ((0, 0, 0, 0), (2, -3, 3, -4), (-6, 6, -12, 12)):
        y, cb, cr = planes(rng, width, height, S, bd, kind)
        data, bs = block_map(rng, width, height)
        a = [p.copy() for p in (y, cb, cr)]
        b = [p.copy() for p in (y, cb, cr)]
        oracle.deblock(a[0], width, a[1], a[2], width // 2, width, height, bd, data, bs, tc2, beta2, cbq, crq)
        reference_c.deblock(b[0], width, b[1], b[2], width // 2, width, height, bd, data, bs, tc2, beta2, cbq, crq)
        for k in range(3):
            assert np.array_equal(a[k], b[k]), (kind, k, tc2)
        if kind == "smooth" and width > 8:
            assert not np.array_equal(a[0], y) and not np.array_equal(a[1], cb)   # the filter did something


@pytest.mark.gpu
@pytest.mark.parametrize("width,height,S,bd", CASES + [(1920, 1080, 1, 8), (640, 360, 2, 10)])
def test_gpu_deblock_equals_the_oracle(oracle, width, height, S, bd):
    from turingcodec_amd import Havoc
    hv = Havoc(0)
    rng = np.random.default_rng(width + 3 * height + bd)
    for kind, (tc2, beta2, cbq, crq) in (("smooth", (0, 0, 0, 0)), ("uniform", (2, -3, 3, -4)), ("extremes", (-6, 6, -12, 12))):
        y, cb, cr = planes(rng, width, height, S, bd, kind)
        data, bs = block_map(rng, width, height)
        exp = [p.copy() for p in (y, cb, cr)]
        oracle.deblock(exp[0], width, exp[1], exp[2], width // 2, width, height, bd, data, bs, tc2, beta2, cbq, crq)
        got = hv.deblock(bd, y.ravel(), width, cb, cr, width // 2, width, height, data, bs, tc2, beta2, cbq, crq)
        assert np.array_equal(got[0].reshape(y.shape), exp[0]), (kind, "Y")
        assert np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]), (kind, "C")
