"""Sample-adaptive offset primitives: the statistics of turing/EncSao.h:62-283 and the filters of turing/sao.cpp.
CPU: oracle/sao_oracle.c against the reference's own code in oracle/_ref and against its committed outputs
(tests/golden/sao_golden.npz).  GPU: havoc_mi355x_sao_stats / _sao_filter against the oracle and the golden outputs, and a
whole-picture run with one job per CTU and colour component."""
import os

import numpy as np
import pytest

import reflibs
import sao_tools

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not os.path.exists(reflibs.REF_SO), reason="oracle/_ref not built (needs the reference sources at build time)")
N = 60      # == tests/golden/make_sao_golden.py


@pytest.fixture(scope="module")
def oracle():
    return reflibs.Oracle()


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "sao_golden.npz"))


def _filter(lib, c, kind):
    d = np.zeros_like(c["rec"])
    lib.sao_filter(d, c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], kind, c["eo_class"], c["band"] if kind == 1 else c["edge"][:5], c["bd"])
    return d


def test_oracle_matches_golden(oracle, golden):
    used = np.zeros(5, bool)
    for seed in range(N):
        c = sao_tools.make_case(seed)
        st = oracle.sao_stats(c["src"], c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], c["bd"])
        assert np.array_equal(st, golden[f"stats{seed}"]), seed
        used |= st[5:10] > 0
        for kind in (1, 2):
            assert np.array_equal(_filter(oracle, c, kind), golden[f"filter{kind}_{seed}"]), (seed, kind)
    assert used.all()      # every edge category occurs


@needs_ref
def test_oracle_matches_reference_on_fresh_cases(oracle):
    ref = reflibs.Reference()
    for seed in range(1000, 1200):
        c = sao_tools.make_case(seed)
        args = (c["src"], c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], c["bd"])
        assert np.array_equal(oracle.sao_stats(*args), ref.sao_stats(*args)), seed
        for kind in (1, 2):
            assert np.array_equal(_filter(oracle, c, kind), _filter(ref, c, kind)), (seed, kind)


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd.havoc import Havoc
    h = Havoc(stream="new")
    yield h
    h.close()


@pytest.mark.gpu
def test_device_matches_golden_and_oracle(hv, oracle, golden):
    from turingcodec_amd.havoc import SAO_JOB_DT
    for seed in list(range(N)) + list(range(2000, 2040)):
        c = sao_tools.make_case(seed)
        jobs = np.array([[c["origin"], c["origin"], c["w"], c["h"]]], np.int32)
        st = hv.sao_stats(c["bd"], c["src"], c["stride"], c["rec"], c["stride"], jobs)[0]
        want = golden[f"stats{seed}"] if seed < N else oracle.sao_stats(c["src"], c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], c["bd"])
        assert np.array_equal(st, want), seed
        for kind in (1, 2):
            j = np.zeros(1, SAO_JOB_DT)
            j["dst_off"], j["src_off"], j["w"], j["h"], j["type"], j["eo_class"] = c["origin"], c["origin"], c["w"], c["h"], kind, c["eo_class"]
            j["offsets"][0] = c["band"] if kind == 1 else c["edge"]
            got = hv.sao_filter(c["bd"], c["rec"], c["stride"], c["rec"], c["stride"], j)
            want = golden[f"filter{kind}_{seed}"] if seed < N else _filter(oracle, c, kind)
            assert np.array_equal(got, want), (seed, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_device_whole_picture_one_job_per_ctu(hv, oracle, bd):
    """416x240 luma + 208x120 chroma planes with a 96 / 48 sample border: statistics and a mix of off / band / edge blocks for every CTU"""
    from turingcodec_amd.havoc import SAO_JOB_DT
    rng = np.random.default_rng(bd)
    dt, mx = (np.uint8, 255) if bd == 8 else (np.uint16, 1023)
    for (W, H, pad, ctu) in ((416, 240, 96, 64), (208, 120, 48, 32)):
        stride = W + 2 * pad
        rec = np.clip(np.kron(rng.integers(0, 14, ((H + 2 * pad) // 8 + 1, stride // 8 + 1)), np.ones((8, 8), int))[:H + 2 * pad, :stride] * (mx // 30) + mx // 4
                      + rng.integers(-3, 4, (H + 2 * pad, stride)), 0, mx).astype(dt).ravel()
        src = np.clip(rec.astype(int) + rng.integers(-5, 6, rec.shape), 0, mx).astype(dt)
        rects = [(x, y, min(ctu, W - x), min(ctu, H - y)) for y in range(0, H, ctu) for x in range(0, W, ctu)]
        sj = np.array([[(y + pad) * stride + x + pad] * 2 + [w, h] for x, y, w, h in rects], np.int32)
        st = hv.sao_stats(bd, src, stride, rec, stride, sj)
        fj = np.zeros(len(rects), SAO_JOB_DT)
        want = np.zeros_like(rec)
        for i, (x, y, w, h) in enumerate(rects):
            o = (y + pad) * stride + x + pad
            assert np.array_equal(st[i], oracle.sao_stats(src, o, stride, rec, o, stride, w, h, bd)), i
            kind, eo = i % 3, int(rng.integers(0, 4))
            offs = np.zeros(32, np.int16)
            offs[:32 if kind == 1 else 5] = rng.integers(-7, 8, 32 if kind == 1 else 5)
            fj[i] = (o, o, w, h, kind, eo, offs, (0, 0))
            oracle.sao_filter(want, o, stride, rec, o, stride, w, h, kind, eo, offs, bd)
        got = hv.sao_filter(bd, rec, stride, rec, stride, fj)
        assert np.array_equal(got, want)


# ---- the joint Cb + Cr band statistics of the chroma SAO decision (EncSao.h:62-109) ------------------------------------
def _chroma_case(seed):
    a, b = sao_tools.make_case(seed), sao_tools.make_case(seed)
    rng = np.random.default_rng(1000 + seed)
    mx = (1 << a["bd"]) - 1
    v_rec = np.clip(a["rec"].astype(np.int64) + rng.integers(-40, 41, a["rec"].shape), 0, mx).astype(a["rec"].dtype)
    v_src = np.clip(v_rec.astype(np.int64) + rng.integers(-6, 7, v_rec.shape), 0, mx).astype(a["rec"].dtype)
    return a, v_src, v_rec


@pytest.mark.parametrize("seed", range(12))
def test_oracle_chroma_band_statistics_equal_the_reference_function(seed, oracle, reference_c):
    a, v_src, v_rec = _chroma_case(seed)
    want = reference_c.sao_band_chroma(a["src"], v_src, a["origin"], a["stride"], a["rec"], v_rec, a["origin"], a["stride"], a["w"], a["h"], a["bd"])
    got = oracle.sao_band_chroma(a["src"], v_src, a["origin"], a["stride"], a["rec"], v_rec, a["origin"], a["stride"], a["w"], a["h"], a["bd"])
    assert np.array_equal(got, want)
    if a["w"] > 2 and a["h"] > 2:
        assert want[32:64].sum() == 2 * (a["w"] - 2) * (a["h"] - 2)


@pytest.mark.gpu
def test_device_chroma_band_statistics_equal_the_oracle(oracle):
    from turingcodec_amd.havoc import Havoc
    hv = Havoc()
    for seed in range(12):
        a, v_src, v_rec = _chroma_case(seed)
        n = a["src"].size
        src = np.concatenate([a["src"], v_src])
        rec = np.concatenate([a["rec"], v_rec])
        jobs = np.array([[a["origin"], n + a["origin"], a["origin"], n + a["origin"], a["w"], a["h"], 0, 0]], np.int32)
        got = hv.sao_band_chroma(a["bd"], src, a["stride"], rec, a["stride"], jobs)[0]
        want = oracle.sao_band_chroma(a["src"], v_src, a["origin"], a["stride"], a["rec"], v_rec, a["origin"], a["stride"], a["w"], a["h"], a["bd"])
        assert np.array_equal(got, want), seed
