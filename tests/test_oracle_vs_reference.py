"""Pins the oracle (oracle/havoc_oracle.c) against the reference's own compiled havoc library
(oracle/_ref/libhavoc_ref.so: C tables = handle 0, x86 JIT tables = handle 1).  CPU only.
Skipped when oracle/_ref has not been built (the golden-vector test then carries the pin)."""
import numpy as np
import pytest

import cases
from cases import PLANE_W as W


def _planes(rng, S, kinds=("uniform", "uniform")):
    bd = 8 if S == 1 else 10
    return [cases.rand_plane(rng, S, bd, kind=k).ravel() for k in kinds]


@pytest.mark.parametrize("S", [1, 2])
@pytest.mark.parametrize("kinds", [("uniform", "uniform"), ("high", "low"), ("extremes", "extremes")])
def test_sad_ssd_satd(oracle, reference_c, reference_jit, S, kinds):
    rng = np.random.default_rng(100 + S)
    a, b = _planes(rng, S, kinds)
    for (w, h, ao, bo) in cases.block_pair_cases(rng, cases.PU_SIZES, 2):
        exp = reference_c.sad(a, ao, W, b, bo, W, w, h)
        assert oracle.sad(a, ao, W, b, bo, W, w, h) == exp
        al = ao - ao % 32  # the JIT loads src with aligned instructions; ref stays unaligned
        assert reference_jit.sad(a, al, W, b, bo, W, w, h) == oracle.sad(a, al, W, b, bo, W, w, h)
    for (w, h, so, ros) in cases.sad4_cases(rng, cases.PU_SIZES, 2):
        exp = reference_c.sad4(a, so, W, b, ros, W, w, h)
        assert oracle.sad4(a, so, W, b, ros, W, w, h) == exp
        al = so - so % 32
        assert reference_jit.sad4(a, al, W, b, ros, W, w, h) == oracle.sad4(a, al, W, b, ros, W, w, h)
    for n in (4, 8, 16, 32, 64):
        for (w, h, ao, bo) in cases.block_pair_cases(rng, [(n, n)], 4):
            exp = reference_c.ssd(a, ao, W, b, bo, W, n, n)
            assert oracle.ssd(a, ao, W, b, bo, W, n, n) == exp
            # the AVX SSD kernel needs 32-byte-aligned rows (havoc/ssd.cpp:122-123): aligned positions only
            ao2, bo2 = (ao // 32) * 32, (bo // 32) * 32
            assert reference_jit.ssd(a, ao2, W, b, bo2, W, n, n) == oracle.ssd(a, ao2, W, b, bo2, W, n, n)
    for n in (2, 4, 8):
        for (w, h, ao, bo) in cases.block_pair_cases(rng, [(n, n)], 8):
            exp = reference_c.satd(a, ao, W, b, bo, W, n)
            assert oracle.satd(a, ao, W, b, bo, W, n) == exp
            ao2, bo2 = (ao // 32) * 32, (bo // 32) * 32
            assert reference_jit.satd(a, ao2, W, b, bo2, W, n) == oracle.satd(a, ao2, W, b, bo2, W, n)


def test_ssd_linear(oracle, reference_c):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, 4096).astype(np.uint8)
    b = rng.integers(0, 256, 4096).astype(np.uint8)
    for n in (1, 16, 100, 512, 4096):
        assert oracle.ssd_linear(a, b, n) == reference_c.ssd_linear(a, b, n)


@pytest.mark.parametrize("S", [1, 2])
def test_pred_uni(oracle, reference_c, reference_jit, S):
    rng = np.random.default_rng(200 + S)
    bds = [8] if S == 1 else [8, 9, 10]
    dt = cases.sample_dtype(S)
    for kind in ("uniform", "extremes"):
        planes = {bd: cases.rand_plane(rng, S, bd, kind=kind).ravel() for bd in bds}
        for (taps, w, h, xf, yf, bd, ro) in cases.pred_uni_cases(rng, bds):
            ref = planes[bd]
            exp = np.zeros(64 * 64 + 64, dt)
            got = np.zeros_like(exp)
            reference_c.pred_uni(exp, 0, 64, ref, ro, W, w, h, xf, yf, bd, taps)
            oracle.pred_uni(got, 0, 64, ref, ro, W, w, h, xf, yf, bd, taps)
            e2 = exp[:64 * 64].reshape(64, 64)[:h, :w]
            g2 = got[:64 * 64].reshape(64, 64)[:h, :w]
            assert np.array_equal(e2, g2), (taps, w, h, xf, yf, bd)
            # the JIT may write to the right of the block (pred_inter.h:27): compare the block only
            j = np.zeros(64 * 64 + 256, dt)
            reference_jit.pred_uni(j, 0, 64, ref, ro, W, w, h, xf, yf, bd, taps)
            assert np.array_equal(j[:64 * 64].reshape(64, 64)[:h, :w], e2), ("jit", taps, w, h, xf, yf, bd)


@pytest.mark.parametrize("S", [1, 2])
def test_pred_bi_subtract_bi(oracle, reference_c, reference_jit, S):
    rng = np.random.default_rng(300 + S)
    bds = [8] if S == 1 else [8, 9, 10]
    dt = cases.sample_dtype(S)
    for kind in ("uniform", "extremes"):
        planes = {bd: cases.rand_plane(rng, S, bd, kind=kind).ravel() for bd in bds}
        for (taps, w, h, xf0, yf0, xf1, yf1, bd, r0, r1) in cases.pred_bi_cases(rng, bds):
            ref = planes[bd]
            exp = np.zeros(64 * 64 + 256, dt)
            got = np.zeros_like(exp)
            jit = np.zeros_like(exp)
            reference_c.pred_bi(exp, 0, 64, ref, r0, r1, W, w, h, xf0, yf0, xf1, yf1, bd, taps)
            oracle.pred_bi(got, 0, 64, ref, r0, r1, W, w, h, xf0, yf0, xf1, yf1, bd, taps)
            reference_jit.pred_bi(jit, 0, 64, ref, r0, r1, W, w, h, xf0, yf0, xf1, yf1, bd, taps)
            e2 = exp[:4096].reshape(64, 64)[:h, :w]
            assert np.array_equal(e2, got[:4096].reshape(64, 64)[:h, :w]), (taps, w, h, xf0, yf0, xf1, yf1, bd)
            assert np.array_equal(e2, jit[:4096].reshape(64, 64)[:h, :w]), ("jit", taps, w, h, bd)
        for bd in bds:
            src = planes[bd]
            pred = cases.rand_plane(rng, S, bd, kind=kind).ravel()
            for (w, h, so, po) in cases.block_pair_cases(rng, cases.PU_SIZES, 1):
                exp = np.zeros(4096, dt)
                got = np.zeros(4096, dt)
                reference_c.subtract_bi(exp, 0, 64, pred, po, W, src, so, W, w, h, bd)
                oracle.subtract_bi(got, 0, 64, pred, po, W, src, so, W, w, h, bd)
                assert np.array_equal(exp, got)


@pytest.mark.parametrize("S", [1, 2])
def test_intra(oracle, reference_c, reference_jit, S):
    rng = np.random.default_rng(400 + S)
    bds = [8] if S == 1 else [8, 9, 10]
    dt = cases.sample_dtype(S)
    for kind in ("uniform", "extremes"):
        for (log2, mode, edge, bd) in cases.intra_cases(bds):
            nb, c = cases.rand_neighbours(rng, S, bd, kind)
            n = 1 << log2
            exp = np.zeros(32 * 32, dt)
            got = np.zeros(32 * 32, dt)
            jit = np.zeros(32 * 32, dt)
            reference_c.intra(exp, 0, 32, nb, c, log2, mode, edge, bd)
            oracle.intra(got, 0, 32, nb, c, log2, mode, 1 if (edge and log2 < 5) else 0, bd)
            reference_jit.intra(jit, 0, 32, nb, c, log2, mode, edge, bd)
            e2 = exp.reshape(32, 32)[:n, :n]
            assert np.array_equal(e2, got.reshape(32, 32)[:n, :n]), (log2, mode, edge, bd)
            assert np.array_equal(e2, jit.reshape(32, 32)[:n, :n]), ("jit", log2, mode, edge, bd)


@pytest.mark.parametrize("bd", [8, 10])
def test_transforms(oracle, reference_c, reference_jit, bd):
    rng = np.random.default_rng(500 + bd)
    S = 1 if bd == 8 else 2
    dt = cases.sample_dtype(S)
    mx = (1 << bd) - 1
    for (log2, tr) in cases.TRANSFORMS:
        n = 1 << log2
        for rep, (lo, hi, kind) in enumerate([(-256, 255, "uniform"), (-mx, mx, "uniform"), (-mx, mx, "extremes"),
                                              (-32768, 32767, "uniform"), (-32768, 32767, "extremes")]):
            res = cases.residual_block(rng, 64, lo, hi, kind).ravel()  # stride 64
            exp = np.zeros(n * n, np.int16)
            got = np.zeros(n * n, np.int16)
            jit = np.zeros(n * n, np.int16)
            reference_c.transform(exp, 0, res, 0, 64, log2, tr, bd)
            oracle.transform(got, 0, res, 0, 64, log2, tr, bd)
            assert np.array_equal(exp, got), ("fwd", log2, tr, rep)
            if rep < 3:  # the AVX2 forward kernels only promise agreement on in-range residuals
                reference_jit.transform(jit, 0, res, 0, 64, log2, tr, bd)
                assert np.array_equal(exp, jit), ("fwd jit", log2, tr, rep)

            # inverse: coefficient ranges from the self-test ([-128,127]) up to full int16
            co = cases.residual_block(rng, n, lo if rep else -128, hi if rep else 127, kind).ravel()
            e16 = np.zeros(n * n, np.int16)
            g16 = np.zeros(n * n, np.int16)
            reference_c.inverse_transform(e16, 0, co, 0, log2, tr, bd)
            oracle.inverse_transform(g16, 0, co, 0, log2, tr, bd)
            assert np.array_equal(e16, g16), ("inv", log2, tr, rep)
            pred = cases.rand_plane(rng, S, bd, 64, 64).ravel()
            ed = np.zeros(64 * 64, dt)
            gd = np.zeros(64 * 64, dt)
            jd = np.zeros(64 * 64, dt)
            reference_c.inverse_transform_add(ed, 0, 64, pred, 0, 64, co, 0, log2, tr, bd)
            oracle.inverse_transform_add(gd, 0, 64, pred, 0, 64, co, 0, log2, tr, bd)
            assert np.array_equal(ed, gd), ("inv add", log2, tr, rep)
            if rep == 0:
                reference_jit.inverse_transform_add(jd, 0, 64, pred, 0, 64, co, 0, log2, tr, bd)
                assert np.array_equal(ed, jd), ("inv add jit", log2, tr)


def test_quantize(oracle, reference_c, reference_jit):
    rng = np.random.default_rng(600)
    for n in (16, 64, 256, 1024):
        src = rng.integers(-32768, 32768, n).astype(np.int16)
        small = rng.integers(-300, 300, n).astype(np.int16)
        # dequant: self-test parameters and real (qp, log2, bitDepth) combinations
        params = [(51, 1), (52224, 1), (51, 4), (52224, 4)]
        for qp in (0, 22, 27, 32, 37, 51):
            for log2 in (2, 3, 4, 5):
                for bd in (8, 10):
                    sc, sh = cases.dequant_params(qp, log2, bd)
                    params.append((sc, sh))
        for (scale, shift) in params:
            for s in (src, small):
                if np.abs(s.astype(np.int64)).max() * scale + (1 << (shift - 1)) >= 2 ** 31:
                    continue  # int overflow is undefined in the reference; never reached by the encoder
                e = np.zeros(n, np.int16)
                g = np.zeros(n, np.int16)
                j = np.zeros(n, np.int16)
                reference_c.quantize_inverse(e, 0, s, 0, scale, shift, n)
                oracle.quantize_inverse(g, 0, s, 0, scale, shift, n)
                reference_jit.quantize_inverse(j, 0, s, 0, scale, shift, n)
                assert np.array_equal(e, g), (scale, shift)
                assert np.array_equal(e, j), ("jit", scale, shift)
        qparams = [(51, 20, 14)]
        for qp in (0, 22, 27, 32, 37, 51):
            for log2 in (2, 3, 4, 5):
                for bd in (8, 10):
                    for intra in (0, 1):
                        sc, sh, of = cases.quant_params(qp, log2, bd, intra)
                        if 16 <= sh <= 27:
                            qparams.append((sc, sh, of))
        for (scale, shift, offset) in qparams:
            for s in (src, small):
                e = np.zeros(n, np.int16)
                g = np.zeros(n, np.int16)
                j = np.zeros(n, np.int16)
                ce = reference_c.quantize(e, 0, s, 0, scale, shift, offset, n)
                cg = oracle.quantize(g, 0, s, 0, scale, shift, offset, n)
                cj = reference_jit.quantize(j, 0, s, 0, scale, shift, offset, n)
                assert np.array_equal(e, g) and (ce != 0) == (cg != 0), (scale, shift, offset)
                assert np.array_equal(e, j) and (ce != 0) == (cj != 0), ("jit", scale, shift, offset)
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        pred = rng.integers(0, 256, 64 * 64).astype(np.uint8)
        res = rng.integers(-300, 300, n * n).astype(np.int16)
        e = np.zeros(64 * 64, np.uint8)
        g = np.zeros(64 * 64, np.uint8)
        reference_c.quantize_reconstruct(e, 0, 64, pred, 0, 64, res, 0, n)
        oracle.quantize_reconstruct(g, 0, 64, pred, 0, 64, res, 0, n)
        assert np.array_equal(e, g)


@pytest.mark.parametrize("S", [1, 2])
def test_pad_block(oracle, reference_c, S):
    """Padding::padBlock (turing/Padding.h:60-97) for every combination of sides; all four sides = numpy edge padding"""
    rng = np.random.default_rng(31)
    dt = cases.sample_dtype(S)
    for (w, h, pad) in ((37, 23, 8), (64, 40, 16), (5, 3, 4), (160, 90, 40)):
        stride, rows = w + 2 * pad + 5, h + 2 * pad
        base = rng.integers(0, 1 << (8 if S == 1 else 10), stride * rows).astype(dt)
        off = pad * stride + pad
        for flags in range(16):
            t, b, l, r = (flags >> 3) & 1, (flags >> 2) & 1, (flags >> 1) & 1, flags & 1
            a1, a2 = base.copy(), base.copy()
            oracle.pad_block(a1, off, w, h, stride, pad, t, b, l, r)
            reference_c.pad_block(a2, off, w, h, stride, pad, t, b, l, r)
            assert np.array_equal(a1, a2), (w, h, pad, flags)
            if flags == 15:
                inner = base.reshape(rows, stride)[pad:pad + h, pad:pad + w]
                assert np.array_equal(a1.reshape(rows, stride)[:, :w + 2 * pad], np.pad(inner, pad, mode="edge"))
                assert np.array_equal(a1.reshape(rows, stride)[:, w + 2 * pad:], base.reshape(rows, stride)[:, w + 2 * pad:])
