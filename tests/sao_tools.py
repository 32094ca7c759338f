"""Seeded inputs for the sample-adaptive-offset tests (tests/test_sao.py, tests/golden/make_sao_golden.py).  Test infrastructure."""
import numpy as np


def make_case(seed):
    """-> dict: padded original / reconstructed planes (flat), stride, origin of the block, w, h, bit depth, filter parameters"""
    rng = np.random.default_rng(seed)
    S = 1 if seed % 2 == 0 else 2
    bd = 8 if S == 1 else int(rng.choice([9, 10]))
    dt = np.uint8 if S == 1 else np.uint16
    w, h = (64, 64) if seed % 5 == 0 else (int(rng.integers(3, 65)), int(rng.integers(3, 65)))
    pad, stride = 2, w + 4 + int(rng.integers(0, 9))
    mx = (1 << bd) - 1
    shape = (h + 2 * pad, stride)
    if seed % 3 == 0:
        rec = rng.integers(0, mx + 1, shape)
    else:      # blocky, smooth content: every edge category and a few bands populated
        rec = np.clip(np.kron(rng.integers(0, 12, (shape[0] // 4 + 1, shape[1] // 4 + 1)), np.ones((4, 4), int))[:shape[0], :shape[1]] * (mx // 40) + mx // 3
                      + rng.integers(-2, 3, shape), 0, mx)
    src = np.clip(rec + rng.integers(-6, 7, shape), 0, mx)
    band = (rng.integers(-7, 8, 32) << (bd - min(bd, 10))).astype(np.int16)
    edge = np.zeros(32, np.int16)
    edge[1:5] = (rng.integers(0, 8, 4) * np.array([1, 1, -1, -1])) << (bd - min(bd, 10))
    return dict(src=src.astype(dt).ravel(), rec=rec.astype(dt).ravel(), stride=stride, origin=pad * stride + pad, w=w, h=h, bd=bd, band=band, edge=edge,
                eo_class=int(rng.integers(0, 4)))
