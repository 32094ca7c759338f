"""havoc_sad_multiref calls BY RUNS (csrc/kernels_metric.hip: k_sad4r, round 5): the consecutive calls of one motion search share their source block and one
staged window.  Runs are an accelerator, never a contract: whatever the run table says, every value equals the reference's per-call result
(havoc/sad.cpp:513-542 restated in oracle/havoc_oracle.c) -- runs as a search makes them (a centre walking, the patterns of turing/Search.hpp:1447-1482),
runs whose box does not fit (far rings), runs cut in the wrong places (calls of different blocks in one run), runs longer than the kernel keeps, single
calls, widths the strips cannot chunk, a window at the very start of the buffer, a base pointer that is not 16-byte aligned; and jobs outside every run are
left alone.  CPU: the run cutter (host code of the C ABI)."""
import numpy as np
import pytest

from test_sad4_window import PATTERNS, SIZES


def search_like_jobs(rng, W, H, stride, pad, sizes, calls_per_run, reach=20, patterns=None, pred=30):
    """runs as a search issues them: same PU, a centre doing a bounded random walk from a predictor, one pattern step per call"""
    pats = [PATTERNS[k] for k in (patterns or ("diamond1", "diamond2", "square4", "hexagon", "ring8a", "ring8b", "bi_grid", "raster", "same"))]
    rows = []
    for (w, h) in sizes:
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        px, py = int(rng.integers(-pred, pred + 1)), int(rng.integers(-pred, pred + 1))      # (pattern reach + walk + predictor stay inside the 96-sample border)
        cx = cy = 0
        so = (y + pad) * stride + x + pad
        for _ in range(calls_per_run):
            cx = int(np.clip(cx + rng.integers(-3, 4), -reach, reach))
            cy = int(np.clip(cy + rng.integers(-3, 4), -reach, reach))
            pat = pats[int(rng.integers(0, len(pats)))]
            rows.append([so] + [(y + py + cy + dy + pad) * stride + x + px + cx + dx + pad for dx, dy in pat] + [w, h, 0])
    return np.array(rows, np.int32)


def check(hv, orc, src, ref, stride, jobs, runs=None, covered=None):
    got = hv.sad4_runs(src, stride, np.ascontiguousarray(ref), stride, jobs, runs)
    bad = []
    for i, j in enumerate(jobs):
        if covered is not None and not covered[i]:
            want = [0, 0, 0, 0]      # outside every run: not computed, the output keeps its zeros
        else:
            want = orc.sad4(src, int(j[0]), stride, ref, [int(v) for v in j[1:5]], stride, int(j[5]), int(j[6]))
        if list(got[i]) != want:
            bad.append((i, list(j), list(got[i]), want))
    return bad


def planes(rng, bit_depth, W, H, stride_extra=0, pad=96):
    dt = np.uint8 if bit_depth == 8 else np.uint16
    stride = W + 2 * pad + stride_extra
    n = (H + 2 * pad) * stride + 64
    return rng.integers(0, 1 << bit_depth, n).astype(dt), rng.integers(0, 1 << bit_depth, n).astype(dt), stride, pad


def test_run_cutter_on_the_host():
    from turingcodec_amd import Havoc
    j = np.zeros((12, 8), np.int32)
    j[:, 0] = [5, 5, 5, 9, 9, 9, 9, 5, 5, 5, 5, 5]
    j[:, 5], j[:, 6] = 16, 16
    j[10:, 6] = 8                                                       # same source offset, another height: another search
    assert Havoc.sad4_make_runs(j, 128)[:, :2].tolist() == [[0, 3], [3, 4], [7, 3], [10, 2]]
    assert Havoc.sad4_make_runs(j, 2)[:, :2].tolist() == [[0, 2], [2, 1], [3, 2], [5, 2], [7, 2], [9, 1], [10, 2]]
    assert Havoc.sad4_make_runs(j[:0]).shape == (0, 8)
    big = np.zeros((300, 8), np.int32)
    big[:, 5], big[:, 6] = 16, 16
    assert Havoc.sad4_make_runs(big, 128)[:, :2].tolist() == [[0, 128], [128, 128], [256, 44]]      # at most 128 calls per run, whatever is asked for
    assert Havoc.sad4_make_runs(big, 1000)[:, :2].tolist() == [[0, 128], [128, 128], [256, 44]]
    assert not Havoc.sad4_make_runs(big, 128)[:, 2:].any()                                             # no stride given: runs without a box
    # with the plane's stride every run carries the box of its candidate blocks, and is cut where the box would outgrow the kernel's window or by block size
    st = 512
    k = np.zeros((40, 8), np.int32)
    k[:, 0], k[:, 5], k[:, 6] = 7, 64, 64
    for i in range(40):
        k[i, 1:5] = [(100 + i) * st + 200 + d for d in (0, 1, -1, 2)]
    r = Havoc.sad4_make_runs(k, 0, st, 1)
    assert r[:, 1].sum() == 40 and r[:, 1].max() <= 16                                                  # 64x64 blocks: 16 calls per run
    for first, count, off, bw, bh in r[:, :5]:
        cand = k[first:first + count, 1:5].ravel()
        assert off == cand.min() - 0 - ((cand.min() % st) - (cand % st).min()) and bw == (cand % st).max() - (cand % st).min() + 64 and bh == cand.max() // st - cand.min() // st + 64
    far = k.copy()
    far[5, 1] += 300 * st                                                                              # one candidate 300 rows away: the box would not fit -> the run is cut there
    rf = Havoc.sad4_make_runs(far, 0, st, 1)
    assert [5, 1] in rf[:, :2].tolist() and rf[rf[:, 0] == 5][0, 3] == 0                                # ... and that call's own run has no box (it cannot have one that fits)


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_every_box_the_cutter_gives_fits_the_kernels_window(bit_depth):
    """host code against kernel arithmetic (no device): k_sad4r stages a run's box at a pitch of 4 * ceil((lead + w S + span_x S) / 16) + 1 dwords per row, lead <= 15 bytes
    of alignment, and takes the fast path only if pitch * rows + 4 <= its window (16 KB x S); a box the cutter thought would fit and does not would silently go call by
    call.  For the picture of the bench and for random search-like tables: every boxed run fits whatever the alignment, holds every candidate block, and is one search."""
    from turingcodec_amd import Havoc
    from turingcodec_amd.workload import FrameWorkload
    S = 1 if bit_depth == 8 else 2
    wl = FrameWorkload(1920, 1080, bit_depth, 11)
    rng = np.random.default_rng(5 + bit_depth)
    tables = [(wl.sad4, wl.stride)]
    src, ref, stride, pad = planes(rng, bit_depth, 320, 256)
    tables.append((search_like_jobs(rng, 320, 256, stride, pad, SIZES * 3, 90, reach=40, pred=40), stride))
    for jobs, st in tables:
        runs = Havoc.sad4_make_runs(jobs, 0, st, S)
        assert runs[:, 1].sum() == len(jobs) and (runs[:, 0] == np.concatenate([[0], np.cumsum(runs[:, 1])[:-1]])).all()      # the runs tile the table in order
        boxed = runs[runs[:, 3] > 0]
        assert len(boxed) > 0.9 * len(runs)
        for first, count, off, bw, bh in boxed[:, :5]:
            j = jobs[first:first + count]
            w, h = int(j[0, 5]), int(j[0, 6])
            assert (j[:, 0] == j[0, 0]).all() and (j[:, 5] == w).all() and (j[:, 6] == h).all()
            chunks = (15 + bw * S + 15) // 16                      # the worst alignment of the box's first sample
            assert (4 * chunks + 1) * bh + 4 <= 4096 * S, (first, count, bw, bh)
            d = j[:, 1:5].astype(np.int64).ravel() - off
            q, r = d // st, d % st
            assert (d >= 0).all() and (q + h <= bh).all() and (r + w <= bw).all(), (first, count)


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth,stride_extra,base_shift", [(8, 0, 0), (8, 5, 3), (10, 0, 0), (10, 3, 5)])
def test_runs_of_a_search_equal_the_oracle(bit_depth, stride_extra, base_shift):
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(70 + bit_depth + stride_extra)
    src, ref, stride, pad = planes(rng, bit_depth, 192, 160, stride_extra)
    if base_shift:
        ref = ref[base_shift:]
    jobs = search_like_jobs(rng, 192, 160, stride, pad, SIZES + [(16, 16)] * 6 + [(8, 8)] * 6, 37)
    runs = Havoc.sad4_make_runs(jobs, 128)                               # runs without a box: the kernel finds each run's box itself
    assert len(runs) >= len(SIZES) and runs[:, 1].max() <= 128
    assert not (bad := check(hv, orc, src, ref, stride, jobs, runs)), bad[:4]
    S = src.itemsize
    boxed = Havoc.sad4_make_runs(jobs, 0, stride, S)                    # the cutter's runs: boxes given, lengths by block size
    assert boxed[:, 3].min() > 0 and boxed[:, 1].sum() == len(jobs)
    assert not (bad := check(hv, orc, src, ref, stride, jobs, boxed)), bad[:4]
    # the same calls cut at odd places, in runs of 1 (boxed and not): same values
    for mr in (5, 1):
        assert not (bad := check(hv, orc, src, ref, stride, jobs, Havoc.sad4_make_runs(jobs, mr))), bad[:4]
        assert not (bad := check(hv, orc, src, ref, stride, jobs, Havoc.sad4_make_runs(jobs, mr, stride, S))), bad[:4]
    # boxes that LIE (too small, shifted, huge): every candidate is checked against the box it was given, a run whose box does not hold goes call by call
    for lie in ("small", "shifted", "huge"):
        wrong = boxed.copy()
        if lie == "small":
            wrong[:, 3] = np.maximum(1, wrong[:, 3] - 3)
            wrong[:, 4] = np.maximum(1, wrong[:, 4] - 2)
        elif lie == "shifted":
            wrong[:, 2] += 2 * stride + 3
        else:
            wrong[:, 3], wrong[:, 4] = stride - 1, 900
        assert not (bad := check(hv, orc, src, ref, stride, jobs, wrong)), (lie, bad[:4])


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_runs_that_cannot_share_a_window_go_call_by_call(bit_depth):
    """far rings (the box of the run does not fit LDS), a run table cut in the wrong places (calls of different blocks and sizes in one run), runs longer
    than the kernel keeps displacements for"""
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(90 + bit_depth)
    src, ref, stride, pad = planes(rng, bit_depth, 256, 192)
    far = search_like_jobs(rng, 256, 192, stride, pad, [(16, 16), (64, 64), (8, 8), (32, 32)], 20, reach=20, patterns=("ring64", "far", "ring16", "diamond1"), pred=8)
    assert not (bad := check(hv, orc, src, ref, stride, far)), bad[:4]
    mixed = search_like_jobs(rng, 256, 192, stride, pad, SIZES, 3)
    wrong = np.array([[0, 7], [7, 50], [57, len(mixed) - 57]], np.int32)      # runs across searches: different source blocks and sizes inside one run
    assert not (bad := check(hv, orc, src, ref, stride, mixed, wrong)), bad[:4]
    long_run = search_like_jobs(rng, 256, 192, stride, pad, [(16, 16)], 300, reach=6)
    assert not (bad := check(hv, orc, src, ref, stride, long_run, np.array([[0, 300]], np.int32))), bad[:4]
    assert not (bad := check(hv, orc, src, ref, stride, long_run, np.array([[0, 128], [128, 128], [256, 44]], np.int32))), bad[:4]


@pytest.mark.gpu
def test_jobs_outside_every_run_are_left_alone_and_bad_runs_are_skipped():
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(12)
    src, ref, stride, pad = planes(rng, 8, 128, 96)
    jobs = search_like_jobs(rng, 128, 96, stride, pad, [(16, 16), (8, 8), (32, 16)], 10)
    runs = np.array([[0, 10], [20, 10], [25, 100], [-3, 5], [5, 0]], np.int32)      # the last three: beyond the table / negative / empty -> skipped
    covered = np.zeros(len(jobs), bool)
    covered[0:10] = covered[20:30] = True
    assert not (bad := check(hv, orc, src, ref, stride, jobs, runs, covered)), bad[:4]


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_generic_widths_and_a_window_at_the_start_of_the_buffer(bit_depth):
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(33 + bit_depth)
    src, ref, stride, pad = planes(rng, bit_depth, 256, 96)
    sizes = [(w, h) for w in (34, 38, 46, 62, 36, 44, 6, 10, 22, 33, 68, 100) for h in (3, 8, 17)]
    jobs = search_like_jobs(rng, 256, 96, stride, pad, sizes, 6, reach=4, patterns=("diamond1", "bi_grid", "same"))
    assert not (bad := check(hv, orc, src, ref, stride, jobs)), bad[:4]
    # a box whose first sample lies in the first 16 bytes of the reference buffer: the aligned copy would start before the buffer -> call by call
    dt = np.uint8 if bit_depth == 8 else np.uint16
    st = 64
    s2, r2 = rng.integers(0, 1 << bit_depth, 64 * st).astype(dt), rng.integers(0, 1 << bit_depth, 64 * st).astype(dt)
    j = np.array([[7, 1, 0, 2, st + 1, 16, 16, 0], [7, 2, 1, 3, st + 2, 16, 16, 0], [7, 17, 16, 18, st + 17, 16, 16, 0]], np.int32)
    assert not (bad := check(hv, orc, s2, r2, st, j)), bad[:4]


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_runs_equal_calls_on_a_picture_of_the_bench(bit_depth):
    """size-independent property at full size: the 1.27 M calls of a 1080p picture by runs == the same calls one by one (k_sad4w), both against the oracle on a sample
    (16-bit samples: a 64-wide block's rows are the 32-dword case of the lane-per-candidate form)"""
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    from turingcodec_amd.workload import FrameWorkload
    hv, orc = Havoc(0), Oracle()
    wl = FrameWorkload(1920, 1080, bit_depth, 11)
    luma, jobs = hv.up(wl.luma), hv.up(wl.sad4)
    runs = Havoc.sad4_make_runs(wl.sad4, 0, wl.stride, wl.S)
    assert 30 < len(wl.sad4) / len(runs) <= 128 and (runs[:, 3] > 0).mean() > 0.99
    a, b = hv.zeros(4 * len(wl.sad4), np.int32), hv.zeros(4 * len(wl.sad4), np.int32)
    hv.sad4_d(luma, wl.stride, luma, wl.stride, jobs, a)
    hv.sad4_runs_d(luma, wl.stride, luma, wl.stride, jobs, hv.up(runs), b)
    a, b = hv.down(a, np.int32).reshape(-1, 4), hv.down(b, np.int32).reshape(-1, 4)
    assert np.array_equal(a, b), np.flatnonzero((a != b).any(axis=1))[:10]
    for i in np.random.default_rng(1).integers(0, len(wl.sad4), 300):
        j = wl.sad4[i]
        assert list(b[i]) == orc.sad4(wl.luma, int(j[0]), wl.stride, wl.luma, [int(v) for v in j[1:5]], wl.stride, int(j[5]), int(j[6]))


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_repeated_candidates_of_a_run(bit_depth):
    """a third of the reference encoder's own havoc_sad_multiref candidates repeat an earlier one of the same search (same source block, size, reference position: counted in
    libhavoc_classic.so over four encodes, NOTEBOOK round 6).  Runs of ONE position asked 512 times, of a handful of positions in every order, of 512 distinct positions, of
    64x64 blocks (row slices), boxes given / found by the kernel: all equal the oracle's per-call values.  (Written for a form of k_sad4r that measured the distinct positions
    once -- an LDS hash table per run -- which was exact and SLOWER, 0.154 -> 0.162 ms: the cases stay.)"""
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    S = 1 if bit_depth == 8 else 2
    rng = np.random.default_rng(400 + bit_depth)
    src, ref, stride, pad = planes(rng, bit_depth, 256, 192)

    def run_of(w, h, offsets):
        x, y = 96, 64
        so = (y + pad) * stride + x + pad
        o = np.asarray(offsets, np.int64).reshape(-1, 4, 2)
        j = np.zeros((len(o), 8), np.int32)
        j[:, 0], j[:, 5], j[:, 6] = so, w, h
        j[:, 1:5] = (y + pad + o[:, :, 1]) * stride + x + pad + o[:, :, 0]
        return j

    same = run_of(16, 16, np.tile([[3, -2]], (128 * 4, 1)))
    few = run_of(16, 16, rng.integers(-2, 3, (128 * 4, 2)))                                             # 25 positions, 512 slots
    grid = np.stack(np.meshgrid(np.arange(-16, 16), np.arange(-8, 8)), -1).reshape(-1, 2)              # 512 distinct positions
    distinct = run_of(8, 8, grid[rng.permutation(512)])
    big = run_of(64, 64, rng.integers(-3, 4, (16 * 4, 2)))
    mixed = run_of(32, 32, np.concatenate([np.tile([[0, 0]], (40, 1)), rng.integers(-20, 21, (152, 2))]))
    for name, jobs in (("same", same), ("few", few), ("distinct", distinct), ("big", big), ("mixed", mixed)):
        for runs in (Havoc.sad4_make_runs(jobs, 128, stride, S), Havoc.sad4_make_runs(jobs, 128)):
            assert not (bad := check(hv, orc, src, ref, stride, jobs, runs)), (name, bad[:4])
