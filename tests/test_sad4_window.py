"""havoc_sad_multiref with the four candidates' common window staged in LDS (csrc/kernels_metric.hip: k_sad4w, round 4) against the oracle
(havoc/sad.cpp:513-542 restated in oracle/havoc_oracle.c) for every way a call can be laid out: the pattern steps of the reference's search
(diamond / star rings / raster line / the bi-directional grid: compact boxes through LDS), candidates too far apart for the box (direct path), every
prediction-unit size incl. the 4-, 12- and 24-wide ones, 8- and 10-bit, strides that are and are not multiples of 16 bytes, a window at the very
start of the buffer, a base pointer that is not 16-byte aligned.  The window form is the default (0.39 against 0.57 ms for a 1080p picture's calls,
csrc/kernels_metric.hip says why); the direct kernel (HAVOC_SAD4_WINDOW=0: two rows in flight, 8 wavefronts per SIMD) and round 1's form
(HAVOC_SAD4_DIRECT=1) go through the same cases."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = [(64, 64), (64, 32), (32, 64), (32, 32), (32, 24), (24, 32), (32, 16), (16, 32), (32, 8), (8, 32), (16, 16), (16, 12), (12, 16), (16, 8), (8, 16), (16, 4), (4, 16),
         (8, 8), (8, 4), (4, 8), (48, 64), (64, 48), (64, 16), (16, 64)]
# pattern steps as Search.hpp issues them (displacements of the four candidates from a centre, full samples)
DIAMOND = [(0, -1), (-1, 0), (1, 0), (0, 1)]
PATTERNS = {"diamond1": [(0, -1), (1, 0), (0, 1), (-1, 0)], "diamond2": [(0, -2), (2, 0), (0, 2), (-2, 0)], "ring8a": [(0, -8), (2, -6), (4, -4), (6, -2)],
            "ring8b": [(8, 0), (6, 2), (4, 4), (2, 6)], "ring16": [(0, -16), (4, -12), (8, -8), (12, -4)], "ring64": [(0, -64), (16, -48), (32, -32), (48, -16)],
            "square4": [(-1, -1), (-1, 1), (1, 1), (1, -1)], "hexagon": [(0, -2), (2, -1), (2, 1), (0, 2)], "raster": [(0, 0), (5, 0), (10, 0), (15, 0)],
            "bi_grid": [(0, 0), (1, 0), (2, 0), (3, 0)], "same": [(3, 2), (3, 2), (3, 2), (3, 2)], "far": [(-40, 30), (50, -45), (7, 60), (-63, -64)]}


def make_jobs(rng, W, H, stride, pad, n_per):
    rows = []
    for (w, h) in SIZES:
        for name, pat in PATTERNS.items():
            for _ in range(n_per):
                x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                cx, cy = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
                so = (y + pad) * stride + x + pad
                ro = [(y + cy + dy + pad) * stride + x + cx + dx + pad for dx, dy in pat]
                rows.append([so] + ro + [w, h, 0])
    return np.array(rows, np.int32)


def run_case(hv, orc, bit_depth, W, H, stride_extra, seed, base_shift=0, n_per=2):
    rng = np.random.default_rng(seed)
    pad = 96
    dt = np.uint8 if bit_depth == 8 else np.uint16
    stride = W + 2 * pad + stride_extra
    rows = H + 2 * pad
    src = rng.integers(0, 1 << bit_depth, rows * stride + 64).astype(dt)
    ref = rng.integers(0, 1 << bit_depth, rows * stride + 64).astype(dt)
    if base_shift:      # a reference base pointer that is not 16-byte aligned: the kernel aligns on the address, not on the offset
        ref = ref[base_shift:]
    jobs = make_jobs(rng, W, H, stride, pad, n_per)
    got = hv.sad4(src, stride, np.ascontiguousarray(ref), stride, jobs)
    bad = []
    for i, j in enumerate(jobs):
        want = orc.sad4(src, int(j[0]), stride, ref, [int(v) for v in j[1:5]], stride, int(j[5]), int(j[6]))
        if list(got[i]) != want:
            bad.append((i, list(j), list(got[i]), want))
    return len(jobs), bad


def _in_a_process_with(env, bit_depth, stride_extra, base_shift, n_per=2):
    """the kernel form is chosen once per process (HAVOC_SAD4_WINDOW / HAVOC_SAD4_DIRECT): run one case in a child with that environment"""
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_sad4_window as t\nfrom reflibs import Oracle\nfrom turingcodec_amd import Havoc\n"
            "n, bad = t.run_case(Havoc(0), Oracle(), %d, 192, 160, %d, %d, %d, %d)\nprint(bad[:3]); print(n, len(bad))\n") % (
                HERE, os.path.dirname(HERE), bit_depth, stride_extra, 11 + bit_depth + stride_extra + base_shift, base_shift, n_per)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    n, nbad = (int(v) for v in out.stdout.split()[-2:])
    return n, nbad, out.stdout[-600:]


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth,stride_extra,base_shift", [(8, 0, 0), (8, 5, 0), (8, 0, 3), (10, 0, 0), (10, 3, 0), (10, 0, 5)])
def test_window_kernel_equals_the_oracle_for_every_layout(bit_depth, stride_extra, base_shift):
    n, nbad, tail = _in_a_process_with({"HAVOC_SAD4_WINDOW": "1"}, bit_depth, stride_extra, base_shift)
    assert n > 500 and nbad == 0, tail


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth,stride_extra,base_shift", [(8, 0, 0), (8, 5, 3), (10, 0, 0), (10, 3, 5)])
def test_default_kernel_equals_the_oracle_for_every_layout(bit_depth, stride_extra, base_shift):
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    n, bad = run_case(hv, orc, bit_depth, 192, 160, stride_extra, 11 + bit_depth + stride_extra + base_shift, base_shift)
    assert n > 500 and not bad, bad[:5]


@pytest.mark.gpu
def test_window_at_the_start_of_the_buffer_and_one_job_launches():
    """a box whose first sample lies in the first 16 bytes of the reference buffer takes the direct path (the aligned copy would start before the buffer)"""
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(5)
    stride = 64
    src = rng.integers(0, 256, 64 * stride).astype(np.uint8)
    ref = rng.integers(0, 256, 64 * stride).astype(np.uint8)
    for ro in ([1, 0, 2, stride + 1], [17, 16, 18, stride + 17], [stride * 3 + 5] * 4):
        for (w, h) in ((16, 16), (8, 8), (32, 16)):
            j = np.array([[7] + ro + [w, h, 0]], np.int32)
            assert list(hv.sad4(src, stride, ref, stride, j)[0]) == orc.sad4(src, 7, stride, ref, ro, stride, w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth,stride_extra,base_shift", [(8, 5, 3), (10, 3, 5)])
def test_direct_kernel_equals_the_oracle(bit_depth, stride_extra, base_shift):
    """HAVOC_SAD4_WINDOW=0: the four blocks read directly (the default until the window form overtook it)"""
    n, nbad, tail = _in_a_process_with({"HAVOC_SAD4_WINDOW": "0"}, bit_depth, stride_extra, base_shift)
    assert n > 500 and nbad == 0, tail


@pytest.mark.gpu
def test_round_one_kernel_is_still_there_and_agrees():
    """HAVOC_SAD4_DIRECT=1: same jobs, same results"""
    n, nbad, tail = _in_a_process_with({"HAVOC_SAD4_DIRECT": "1"}, 8, 0, 0, 1)
    assert n > 200 and nbad == 0, tail


@pytest.mark.gpu
@pytest.mark.parametrize("bit_depth", [8, 10])
def test_generic_widths_with_nearby_candidates(bit_depth):
    """widths the strips cannot cover with 16 chunks of the size their alignment picks (16-bit 34, 38, 46, 62: row bytes % 8 == 4
    beyond 64 bytes; 8-bit 68 ... 128) -- with NEARBY candidates, so that the window condition holds but for the width: they must take the direct / generic
    path, never return the zero sums a zero-rows-per-iteration strip would"""
    from reflibs import Oracle
    from turingcodec_amd import Havoc
    hv, orc = Havoc(0), Oracle()
    rng = np.random.default_rng(40 + bit_depth)
    dt = np.uint8 if bit_depth == 8 else np.uint16
    pad, W, H = 96, 256, 96
    stride = W + 2 * pad
    src = rng.integers(0, 1 << bit_depth, (H + 2 * pad) * stride + 64).astype(dt)
    ref = rng.integers(0, 1 << bit_depth, (H + 2 * pad) * stride + 64).astype(dt)
    widths = [34, 38, 42, 46, 50, 54, 58, 62, 36, 44, 52, 60, 33, 35, 6, 10, 14, 18, 22, 26, 30, 68, 72, 76, 100, 128]
    rows = []
    for w in widths:
        for h in (1, 3, 8, 17, 32):
            for pat in (PATTERNS["diamond1"], PATTERNS["ring8a"], PATTERNS["bi_grid"], PATTERNS["same"]):
                x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                so = (y + pad) * stride + x + pad
                rows.append([so] + [(y + dy + pad) * stride + x + dx + pad for dx, dy in pat] + [w, h, 0])
    jobs = np.array(rows, np.int32)
    got = hv.sad4(src, stride, ref, stride, jobs)
    bad = [(list(j), list(got[i])) for i, j in enumerate(jobs)
           if list(got[i]) != orc.sad4(src, int(j[0]), stride, ref, [int(v) for v in j[1:5]], stride, int(j[5]), int(j[6]))]
    assert not bad, bad[:5]
