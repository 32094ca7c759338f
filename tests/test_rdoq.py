"""Rate-distortion optimised quantisation: turing/Rdoq.cpp:37-1023.

CPU (-m "not gpu"): the C restatement oracle/rdoq_oracle.c against (a) the reference's own Rdoq.cpp / ScanOrder.cpp / Cabac.cpp
compiled into oracle/_ref where that library exists and (b) the committed outputs of that library (tests/golden/rdoq_golden.npz,
made by tests/golden/make_rdoq_golden.py).  GPU (-m gpu): havoc_mi355x_rdoq through the C ABI against the same oracle and golden
data -- bit exact, every level and every coded-block flag.
"""
import os

import numpy as np
import pytest

import rdoq_tools as rt
import reflibs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "rdoq_golden.npz")
HAVE_REF = os.path.exists(reflibs.REF_SO)
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built (needs the reference sources at build time)")
CASES = [(2, 8, 300), (3, 8, 200), (4, 8, 100), (5, 8, 60), (2, 10, 100), (3, 10, 100), (4, 10, 60), (5, 10, 40)]   # == make_rdoq_golden.py


@pytest.fixture(scope="module")
def oracle():
    return reflibs.Oracle()


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _golden_case(golden, log2, bd, count):
    src, states, blocks = rt.make_blocks(1000 + 10 * log2 + bd, log2, bd, count, initial_states=list(golden["initial_states"]))
    k = f"l{log2}b{bd}"
    crc = np.bitwise_xor.reduce(src.view(np.uint16).astype(np.uint32) * np.arange(1, len(src) + 1, dtype=np.uint32))
    assert crc == golden[k + ".src_crc"][0], "the generator no longer reproduces the golden inputs"
    return src, states, blocks, golden[k + ".levels"], golden[k + ".cbf"]


@needs_ref
def test_scan_tables_equal_the_reference(oracle):
    ref = reflibs.Reference()
    for log2 in range(1, 6):
        for scan in range(3):
            for pos in range(1 << 2 * log2):
                for comp in range(2):
                    assert oracle.scan_order(log2, scan, pos, comp) == ref.scan_order(log2, scan, pos, comp)


@pytest.mark.parametrize("log2,bd,count", CASES)
def test_oracle_matches_golden(oracle, golden, log2, bd, count):
    src, states, blocks, levels, cbf = _golden_case(golden, log2, bd, count)
    got, got_cbf = rt.run_cpu(oracle, src, states, blocks)
    assert np.array_equal(got, levels) and np.array_equal(got_cbf, cbf)
    assert (levels != 0).any() and (cbf == 0).any() and (cbf != 0).any()


@needs_ref
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_oracle_matches_reference_on_fresh_blocks(oracle, log2):
    """seeds the golden file has not seen, 8- / 10- / 12-bit"""
    ref = reflibs.Reference()
    init = [ref.rdoq_initial_states(qp, t) for qp, t in ((27, 0), (30, 2))]
    for bd in (8, 10, 12):
        src, states, blocks = rt.make_blocks(77 + log2 + bd, log2, bd, 1200 >> (log2 - 2), initial_states=init)
        a, ca = rt.run_cpu(oracle, src, states, blocks)
        b, cb = rt.run_cpu(ref, src, states, blocks)
        assert np.array_equal(a, b) and np.array_equal(ca, cb)


def test_rdoq_changes_levels_the_plain_quantiser_keeps(oracle):
    """sanity of the inputs: the optimiser must actually drop / lower levels and hide signs on a good share of the blocks"""
    src, states, blocks = rt.make_blocks(5, 3, 8, 300)
    got, _ = rt.run_cpu(oracle, src, states, blocks)
    changed = hidden = 0
    for b in blocks:
        o = b["src_off"]
        s = src[o:o + 64].astype(np.int64)
        plain = np.sign(s) * ((np.abs(s) * b["quant_scale"] + (1 << (b["quant_shift"] - 1))) >> b["quant_shift"])
        changed += int((plain != got[o:o + 64]).any())
        hidden += int((np.abs(got[o:o + 64]) > np.abs(plain)).any())
    assert changed > 150 and hidden > 10


def test_lambda_helper_of_the_library_equals_the_constructor(oracle):
    from turingcodec_amd import havoc
    rng = np.random.default_rng(3)
    for _ in range(500):
        lam = float(rng.uniform(0.05, 4000))
        inv = int(rng.choice(rt.INV_SCALE)) << int(rng.integers(0, 9))
        assert havoc.rdoq_lambda(lam, inv) == oracle.rdoq_lambda(lam, inv)


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd.havoc import Havoc
    h = Havoc(stream="new")
    yield h
    h.close()


def _device(hv, src, states, blocks):
    from turingcodec_amd import havoc
    b0 = blocks[0]
    return hv.rdoq(b0["bit_depth"], b0["log2"], src, states, rt.device_jobs(blocks, havoc.rdoq_lambda))


def _explain(src, blocks, got, want):
    bad = [i for i, b in enumerate(blocks) if not np.array_equal(got[b["src_off"]:b["src_off"] + (1 << 2 * b["log2"])], want[b["src_off"]:b["src_off"] + (1 << 2 * b["log2"])])]
    b = blocks[bad[0]]
    o, n2 = b["src_off"], 1 << 2 * b["log2"]
    d = np.flatnonzero(got[o:o + n2] != want[o:o + n2])
    return f"{len(bad)} of {len(blocks)} blocks differ; first {bad[0]}: {b}; positions {d[:8]}, got {got[o:o + n2][d[:8]]}, want {want[o:o + n2][d[:8]]}"


@pytest.mark.gpu
@pytest.mark.parametrize("log2,bd,count", CASES)
def test_device_matches_golden(hv, golden, log2, bd, count):
    src, states, blocks, levels, cbf = _golden_case(golden, log2, bd, count)
    got, got_cbf = _device(hv, src, states, blocks)
    assert np.array_equal(got, levels), _explain(src, blocks, got, levels)
    assert np.array_equal(got_cbf, cbf)


@pytest.mark.gpu
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_device_matches_oracle_on_many_blocks(hv, oracle, log2, bd):
    """block counts that do not fill the last workgroup; several state snapshots in one launch"""
    count = (4099 >> (log2 - 2)) + 3
    src, states, blocks = rt.make_blocks(900 + log2 * 16 + bd, log2, bd, count, n_states=11)
    want, want_cbf = rt.run_cpu(oracle, src, states, blocks)
    got, got_cbf = _device(hv, src, states, blocks)
    assert np.array_equal(got, want), _explain(src, blocks, got, want)
    assert np.array_equal(got_cbf, want_cbf)


@pytest.mark.gpu
def test_device_edge_cases(hv, oracle):
    """no jobs; one job; blocks scattered in a larger buffer with dst != src offsets"""
    from turingcodec_amd import havoc
    src, states, blocks = rt.make_blocks(31, 4, 8, 5)
    jobs = rt.device_jobs(blocks, havoc.rdoq_lambda)
    got, cbf = hv.rdoq(8, 4, src, states, jobs[:0])
    assert not got.any() and len(cbf) == 0
    want, want_cbf = rt.run_cpu(oracle, src, states, blocks)
    got, cbf = hv.rdoq(8, 4, src, states, jobs[:1])
    assert np.array_equal(got[:256], want[:256]) and not got[256:].any() and cbf[0] == want_cbf[0]
    # a workspace smaller than havoc_mi355x_rdoq_workspace(njobs), or levels written over the coefficients, is refused
    import torch
    from turingcodec_amd.havoc import HavocError
    d_src = hv.up(src)
    d_dst = hv.zeros(len(src), np.int16)
    d_states = torch.from_numpy(states.reshape(-1)).to(hv.device)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(hv.device)
    d_cbf = hv.zeros(len(jobs), np.int32)
    with pytest.raises(HavocError):
        hv.rdoq_d(8, 4, d_dst, d_src, d_states, d_jobs, d_cbf, torch.zeros(4, dtype=torch.int64, device=hv.device))
    with pytest.raises(HavocError):
        hv.rdoq_d(8, 4, d_src, d_src, d_states, d_jobs, d_cbf, hv.rdoq_workspace(len(jobs)))
    jobs["dst_off"] = jobs["src_off"][::-1]
    got, cbf = hv.rdoq(8, 4, src, states, jobs)
    for i, b in enumerate(blocks):
        o, d = b["src_off"], int(jobs["dst_off"][i])
        assert np.array_equal(got[d:d + 256], want[o:o + 256])
    assert np.array_equal(cbf, want_cbf)


_WALK_FORMS_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import rdoq_tools as rt, reflibs
from turingcodec_amd import havoc
hv = havoc.Havoc()
oracle = reflibs.Oracle()
for seed, count in ((77, 37), (78, 130)):
    src, states, blocks = rt.make_blocks(seed, 5, 8, count, n_states=5)
    want, want_cbf = rt.run_cpu(oracle, src, states, blocks)
    got, got_cbf = hv.rdoq(8, 5, src, states, rt.device_jobs(blocks, havoc.rdoq_lambda))
    assert np.array_equal(got, want) and np.array_equal(got_cbf, want_cbf), (seed, int((got != want).sum()))
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("sort", ["0", "1"])
@pytest.mark.parametrize("form", ["0", "4", "8"])
def test_every_walk_form_of_32x32_blocks_matches_the_oracle(form, sort):
    """HAVOC_RDOQ_SORT (round 6; default 1): 0 = a launch this small is walked in job order (no histogram, no counting sort; the diagonal walk's last workgroups take the
    blocks of the other scans), 1 = sorted densest-first as a 4K picture's launch is.  HAVOC_RDOQ_DIAG picks how 32x32 blocks are walked -- 0: sequential kernel only, 4 / 8: the diagonal walk with that many lanes per
    block (0 is the default since round 6: the step is bound by instruction issue and the diagonal walk issues 3.5x the instructions) -- and is read once per process, so each form runs in its own interpreter; diagonal-scan blocks and
    the few horizontal / vertical ones of make_blocks (which go to the sequential kernel whatever the form) in one launch"""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _WALK_FORMS_SCRIPT, ROOT], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HAVOC_RDOQ_DIAG=form, HAVOC_RDOQ_SORT=sort, HAVOC_RDOQ_TINY="0"))      # (TINY 0: launches this small would otherwise be one kernel in job order)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


_WALK_FORMS_16_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import rdoq_tools as rt, reflibs
from turingcodec_amd import havoc
hv = havoc.Havoc()
oracle = reflibs.Oracle()
for bd, seed, count in ((8, 81, 50), (8, 82, 700), (10, 83, 300)):
    src, states, blocks = rt.make_blocks(seed, 4, bd, count, n_states=5)
    want, want_cbf = rt.run_cpu(oracle, src, states, blocks)
    got, got_cbf = hv.rdoq(bd, 4, src, states, rt.device_jobs(blocks, havoc.rdoq_lambda))
    assert np.array_equal(got, want) and np.array_equal(got_cbf, want_cbf), (seed, int((got != want).sum()))
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("sort", ["0", "1"])
@pytest.mark.parametrize("form", ["0", "4", "8"])
def test_every_walk_form_of_16x16_blocks_matches_the_oracle(form, sort):
    """(HAVOC_RDOQ_SORT as above.)  HAVOC_RDOQ_DIAG16: how 16x16 blocks are walked -- 0 (default): sequential kernel (a lane per block), 4 / 8: the
    anti-diagonal walk with that many lanes per block; each form in its own interpreter (the switch is read once per process)"""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _WALK_FORMS_16_SCRIPT, ROOT], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HAVOC_RDOQ_DIAG16=form, HAVOC_RDOQ_SORT=sort, HAVOC_RDOQ_TINY="0"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]
