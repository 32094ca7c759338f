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


# ---- boundary strengths derived from the block structure (havoc_mi355x_derive_bs) -------------------------------------
PART = [[(0, 0, 4, 4)], [(0, 0, 4, 2), (0, 2, 4, 2)], [(0, 0, 2, 4), (2, 0, 2, 4)], [(0, 0, 4, 1), (0, 1, 4, 3)], [(0, 0, 4, 3), (0, 3, 4, 1)],
        [(0, 0, 1, 4), (1, 0, 3, 4)], [(0, 0, 3, 4), (3, 0, 1, 4)]]      # part modes in quarters of the unit size


def random_structure(rng, W, H, p_intra=0.25):
    """a random but legal block structure: coding quadtree down to 8x8, inter part modes with uni / bi motion from a few reference pictures
    (vectors clustered so that some neighbours match, some differ by less than a sample, some by more), transform trees one or two levels
    deep with random coded flags, a few bypass units, per-unit QP.  Returns unit lists (for the reference's event functions) and the same
    structure rasterised into 4x4 cells (for the oracle / device)."""
    from turingcodec_amd.havoc import CELL_DT, CELL_INTRA, CELL_CODED, CELL_NO_FILTER, CELL_PU_LEFT, CELL_PU_TOP
    cus, pus, tus = [], [], []
    cells = np.zeros((H // 4, W // 4), CELL_DT)
    cells["dpb_index"] = -1
    base = [np.array([12, 8]), np.array([-20, 4]), np.array([0, 0])]

    def motion():
        m = base[int(rng.integers(0, 3))] + rng.integers(-5, 6, 2)
        return int(m[0]), int(m[1])

    def tu(x, y, log2, intra, depth):
        if log2 > 2 and (log2 > 5 or (depth < 2 and rng.random() < 0.4)):
            for k in range(4):
                tu(x + (k & 1) * (1 << (log2 - 1)), y + (k >> 1) * (1 << (log2 - 1)), log2 - 1, intra, depth + 1)
            return
        cbf = int(rng.random() < 0.5)
        tus.append((x, y, log2, cbf, intra))
        c = cells[y // 4:(y + (1 << log2)) // 4, x // 4:(x + (1 << log2)) // 4]
        c["tu_log2"] = log2
        c["flags"] |= (CELL_CODED if cbf else 0)

    def cu(x, y, log2):
        if x >= W or y >= H:
            return
        size = 1 << log2
        if log2 > 3 and (x + size > W or y + size > H or rng.random() < (0.9 if log2 == 6 else 0.5)):
            for k in range(4):
                cu(x + (k & 1) * size // 2, y + (k >> 1) * size // 2, log2 - 1)
            return
        intra = int(rng.random() < p_intra)
        qp, bypass = int(rng.integers(20, 40)), int(rng.random() < 0.05)
        cus.append((x, y, log2, intra, qp, bypass))
        c = cells[y // 4:(y + size) // 4, x // 4:(x + size) // 4]
        c["qp_y"] = qp
        c["flags"] = (CELL_INTRA if intra else 0) | (CELL_NO_FILTER if bypass else 0)
        if not intra:
            modes = PART[:3] if log2 == 3 else PART
            q = size // 4
            for dx, dy, w4, h4 in modes[int(rng.integers(0, len(modes)))]:
                px, py, pw, ph = x + dx * q, y + dy * q, w4 * q, h4 * q
                kind = int(rng.integers(0, 3))      # list 0, list 1, both
                m0, m1 = motion(), motion()
                d0 = int(rng.integers(0, 3)) if kind != 1 else -1
                d1 = int(rng.integers(0, 3)) if kind != 0 else -1
                pus.append((px, py, pw, ph, m0[0], m0[1], m1[0], m1[1], d0, d1))
                pc = cells[py // 4:(py + ph) // 4, px // 4:(px + pw) // 4]
                pc["mv"][..., 0, :] = m0
                pc["mv"][..., 1, :] = m1
                pc["dpb_index"][..., 0] = d0
                pc["dpb_index"][..., 1] = d1
                pc["flags"][:, 0] |= CELL_PU_LEFT
                pc["flags"][0, :] |= CELL_PU_TOP
        tu(x, y, log2, intra, 0)

    for y in range(0, H, 64):
        for x in range(0, W, 64):
            cu(x, y, 6)
    return np.array(cus, np.int32), np.array(pus, np.int32).reshape(-1, 10), np.array(tus, np.int32), cells


@pytest.mark.parametrize("W,H,seed", [(128, 64, 1), (416, 240, 2), (200, 136, 3), (640, 360, 4)])
def test_oracle_boundary_strengths_equal_the_reference_derivation(oracle, reference_c, W, H, seed):
    """per-cell restatement (oracle_derive_bs) vs the reference's own processCu / processTu / processRc + sameMotion over the units"""
    cus, pus, tus, cells = random_structure(np.random.default_rng(seed), W, H)
    want = reference_c.derive_bs(W, H, cus, pus, tus)
    got = oracle.derive_bs(cells, W, H)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert len(pus) > 10 and (want[1] != 0).mean() > 0.3 and len(np.unique(want[1] & 3)) == 3      # strengths 0, 1 and 2 all occur


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,seed", [(128, 64, 1), (416, 240, 2), (1920, 1080, 5)])
def test_device_boundary_strengths_equal_the_oracle_and_feed_the_filter(oracle, W, H, seed):
    from turingcodec_amd.havoc import Havoc
    rng = np.random.default_rng(seed)
    cus, pus, tus, cells = random_structure(rng, W, H)
    hv = Havoc()
    want = oracle.derive_bs(cells, W, H)
    got = hv.derive_bs(cells, W, H)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # ... and the derived arrays through the filter: device == oracle
    y, cb, cr = planes(rng, W, H, 1, 8, "smooth")
    ref = [p.copy() for p in (y, cb, cr)]
    oracle.deblock(ref[0], W, ref[1], ref[2], W // 2, W, H, 8, want[0], want[1])
    have = hv.deblock(8, y.ravel(), W, cb, cr, W // 2, W, H, got[0], got[1])
    assert all(np.array_equal(np.asarray(h).reshape(w.shape), w) for h, w in zip(have, ref))
    assert not np.array_equal(ref[0], y)
