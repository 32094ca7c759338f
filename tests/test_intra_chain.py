"""An intra picture with the REAL dependencies between its partitions (turingcodec_amd.decisions.IntraChainPicture, csrc/kernels_decide.hip: k_intra_gather /
k_intra_commit) against the block-at-a-time loop the reference runs: partitions in coding order, each one's 4n + 1 reference samples taken from
the RUNNING reconstruction with the substitution process of HEVC 8.4.4.2.2 (turing/Reconstruct.cpp:609-615, IntraReferenceSamples.h), candModeList from the modes
decided left of and above it (turing/CandModeList.h:33-95), then the 35-mode SATD stage, the refinement order and the RD refinement of every candidate through the
reference's own intra / Hadamard / transform tables and Rdoq.cpp (tests/search_client.cpp, the "ref" arm), the champion's reconstruction written back before the next
partition starts.  Champions, costs, candModeLists and the whole reconstructed picture must be identical."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def test_partitions_tile_the_picture_and_levels_respect_every_dependency():
    from turingcodec_amd import workload
    for w, h in ((416, 240), (640, 360)):
        parts, owner, level = workload.intra_picture_partitions(w, h, 5)
        area = sum(1 << (2 * int(q["log2"])) for q in parts)
        assert area == w * h and (owner >= 0).all()
        assert set(np.unique(parts["log2"])) <= {2, 3, 4, 5} and len(np.unique(parts["log2"])) == 4
        # coding order: CTUs in raster order, z-order inside (a partition's owner cells are written once, in increasing index)
        ctu = (parts["y0"] // 64) * ((w + 63) // 64) + parts["x0"] // 64
        assert (np.diff(ctu) >= 0).all()
        for i, q in enumerate(parts):      # every sample a partition may read belongs to an EARLIER level (or is not available to it)
            n = 1 << int(q["log2"])
            for k in range(4 * n + 1):
                x = q["x0"] - 1 if k <= 2 * n else q["x0"] + k - 2 * n - 1
                y = q["y0"] + 2 * n - 1 - k if k < 2 * n else q["y0"] - 1
                if 0 <= x < w and 0 <= y < h and owner[y >> 2, x >> 2] < i:
                    assert level[owner[y >> 2, x >> 2]] < level[i]
        assert level.min() == 0 and (np.bincount(level) > 0).all()


def _reference_samples(rec2d, owner, pad, w, h, q, i, bit_depth):
    """HEVC 8.4.4.2.2 on the running reconstruction: (unfiltered, filtered) arrays of 4n + 1 samples, left column from the bottom, corner, row above"""
    n = 1 << int(q["log2"])
    k = np.arange(4 * n + 1)
    x = np.where(k <= 2 * n, q["x0"] - 1, q["x0"] + k - 2 * n - 1)
    y = np.where(k < 2 * n, q["y0"] + 2 * n - 1 - k, q["y0"] - 1)
    inside = (x >= 0) & (y >= 0) & (x < w) & (y < h)
    have = inside.copy()
    have[inside] = owner[y[inside] >> 2, x[inside] >> 2] < i
    val = np.where(have, rec2d[np.clip(y, 0, h - 1) + pad, np.clip(x, 0, w - 1) + pad].astype(np.int64), 0)
    if not have.any():
        val[:] = 1 << (bit_depth - 1)
    else:
        first = int(np.argmax(have))
        val[0] = val[first]
        for j in range(1, len(val)):
            if not have[j]:
                val[j] = val[j - 1]
    # the filtered copy: IntraReferenceSamples::filter as restated in the oracle -- pinned against the arrays the reference ENCODER held (tests/trace_runner.py:
    # intra_neighbours, 0 of 3 230 differ), strong intra smoothing of flat 32x32 edges included (the encoder's default, Encoder.cpp:688)
    from reflibs import Oracle
    global _ORC
    try:
        orc = _ORC
    except NameError:
        orc = _ORC = Oracle()
    filt = orc.intra_filter_neighbours(val.astype(np.int32), n, bit_depth, 1).astype(np.int64)
    return val, filt


def _cand_mode_list(modes, owner, w, h, q, i):
    def there(x, y):
        return 0 <= x < w and 0 <= y < h and owner[y >> 2, x >> 2] < i
    a = int(modes[q["y0"] >> 2, (q["x0"] - 1) >> 2]) if there(q["x0"] - 1, q["y0"]) else 1
    b = int(modes[(q["y0"] - 1) >> 2, q["x0"] >> 2]) if there(q["x0"], q["y0"] - 1) and q["y0"] - 1 >= (q["y0"] >> 6) << 6 else 1
    if a == b:
        return ([0, 1, 26] if a < 2 else [a, ((a + 29) % 32) + 2, ((a - 1) % 32) + 2]), 1
    return [a, b, 0 if (a != 0 and b != 0) else (1 if (a != 1 and b != 1) else 26)], 2


def _host_chain(ref, ip):
    """the reference's loop, one partition at a time"""
    from turingcodec_amd import workload
    from turingcodec_amd.decisions import INTRA_RD_RESULT_DT
    w, h, pad, stride, BD = ip.W, ip.H, ip.PAD, ip.stride, ip.bd
    rec = np.zeros_like(ip.host_src)
    rec2d = rec.reshape(-1, stride)
    modes = np.zeros(ip.owner.shape, np.int32)
    best = np.zeros(len(ip.parts), INTRA_RD_RESULT_DT)
    cands = np.zeros((len(ip.parts), 3), np.int32)
    where = {log2: {int(p): k for k, p in enumerate(g["sel"])} for log2, g in ip.sizes.items()}
    for i, q in enumerate(ip.parts):
        log2 = int(q["log2"])
        n, g = 1 << log2, ip.sizes[log2]
        k = where[log2][i]
        nbu, nbf = _reference_samples(rec2d, ip.owner, pad, w, h, q, i, BD)
        nb = np.concatenate([nbu, nbf]).astype(ip.dt)
        job = g["jobs"][k:k + 1].copy()
        job[0, 1], job[0, 2] = 2 * n + 1, 4 * n + 1 + 2 * n + 1
        ictx = g["ictx"][k:k + 1].copy()
        ictx["cand_mode_list"][0], ictx["neighbour_modes"][0] = _cand_mode_list(modes, ip.owner, w, h, q, i)
        cands[i] = ictx["cand_mode_list"][0]
        order = ref.intra_order(ictx, ip.rsl, ref.intra35(BD, log2, ip.host_src, stride, nb, job))
        b, r = ref.intra_rd(BD, log2, ip.host_src, stride, nb, job, order, ictx, g["ctu"][k:k + 1], ip.rdoq_states, ip.quant[log2 - 2], ip.lam, 1.0 / ip.lam)
        best[i] = b[0]
        rec2d[q["y0"] + pad:q["y0"] + pad + n, q["x0"] + pad:q["x0"] + pad + n] = r[0].reshape(n, n)
        modes[q["y0"] >> 2:(q["y0"] + n) >> 2, q["x0"] >> 2:(q["x0"] + n) >> 2] = b[0]["mode"]
    return best, cands, rec


@pytest.mark.parametrize("res,BD,qp", [("416x240", 8, 32), ("640x360", 10, 27)])
def test_chain_client_on_the_cpu_stand_in_device_equals_the_reference_loop(res, BD, qp):
    """havoc_search_intra_chain (the level loop, the per-size slices, the worst-case candidate slots) over tests/mock_device.c, where the chain's device-side steps --
    gather with substitution, candModeList, commit, spare slots -- are restated on the host: champions, candModeLists and the reconstruction equal the reference's loop"""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "intra_chain_runner.py"), "--device", "mock", "--res", res, "--bit-depth", str(BD), "--qp", str(qp)],
                         capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["partitions"] > 1000 and r["levels"] > 100 and r["launches"] > 13 * r["levels"]
    assert r["mismatching_champions"] == 0 and r["mismatching_cand_mode_lists"] == 0 and r["reconstruction_equal"], r
    assert r["distinct_champion_modes"] > 10 and r["coded"] > 0.05, r


@pytest.mark.gpu
def test_both_forms_of_the_level_loop_agree():
    """a wait per level (havoc_search_intra_device per level: candidate counts known to the host) == no wait at all (worst-case candidate slots, spare ones filled)"""
    from turingcodec_amd.decisions import IntraChainPicture
    from turingcodec_amd.havoc import Havoc
    hv = Havoc(stream="new")
    ip = IntraChainPicture(hv, 416, 240, 8, 32, seed=29)
    ip.step(wait_per_level=True)
    a = ip.results()
    import torch
    with torch.cuda.stream(hv.tstream):
        ip.d_rec.zero_()
        ip.d_modes.zero_()
    ip.step()
    b = ip.results()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("res,BD,qp", [((416, 240), 8, 32), ((640, 360), 10, 27), ((1920, 1080), 8, 32)])
def test_intra_picture_with_running_reconstruction_equals_the_reference_loop(res, BD, qp):
    import search_tools as st
    from turingcodec_amd.decisions import IntraChainPicture
    from turingcodec_amd.havoc import Havoc
    hv = Havoc(stream="new")
    ip = IntraChainPicture(hv, res[0], res[1], BD, qp, seed=17)
    import time
    ip.step()
    t0 = time.perf_counter()
    ip.step()
    seconds = time.perf_counter() - t0
    best, cand, nbm, rec = ip.results()
    ref = st.Client("ref", 3)
    exp_best, exp_cand, exp_rec = _host_chain(ref, ip)
    assert np.array_equal(cand, exp_cand)
    assert best.tobytes() == exp_best.tobytes()
    pad, stride = ip.PAD, ip.stride
    got2d, exp2d = rec.reshape(-1, stride), exp_rec.reshape(-1, stride)
    assert np.array_equal(got2d[pad:pad + res[1], pad:pad + res[0]], exp2d[pad:pad + res[1], pad:pad + res[0]])
    modes = best["mode"]
    assert len(np.unique(modes)) > 10 and (best["outcome"]["cbf"] != 0).any()
    # the dependencies matter: the same partitions predicted from the SOURCE picture's neighbours decide differently somewhere
    print(res, BD, qp, "partitions", len(ip.parts), "levels", ip.nlevels, "launches", ip.launches, "seconds per picture", round(seconds, 4))
