"""Full-size GPU checks through size-independent properties (the oracle would take minutes at these sizes):
the bench workload at BASELINE.json's resolutions must give identical results on independent code paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd import Havoc
    return Havoc(0, stream="new")


def _frames(hv, res, bit_depth, seed):
    import bench
    from turingcodec_amd.workload import FrameWorkload
    w, h = res
    wl = FrameWorkload(w, h, bit_depth, seed)
    return wl, bench.DeviceFrame(hv, wl, use_planes=True), bench.DeviceFrame(hv, wl, use_planes=False)


@pytest.mark.parametrize("res,bit_depth", [((1920, 1080), 8), ((3840, 2160), 8), ((1920, 1080), 10)])
def test_subpel_planes_equal_fused_candidates(hv, res, bit_depth):
    """every sub-pel candidate cost is the same whether it is measured against the precomputed phase planes
    (interp_planes + satd_multi) or by the fused per-candidate kernel (subpel_satd): two independent implementations of
    costDistortionMv, ~184 k (1080p) / ~737 k (4K) candidates"""
    wl, a, b = _frames(hv, res, bit_depth, 5)
    a.step()
    b.step()
    hv.sync()
    n = sum(len(v) for v in wl.subpel_idx.values())
    ca, cb = np.zeros(n, np.int32), np.zeros(n, np.int32)
    for c, g in a.subpel_planes.items():
        idx = wl.subpel_planes_idx[c].ravel()
        ca[idx[idx >= 0]] = hv.down(g["cost"], np.int32)[idx >= 0]
    for c, g in b.subpel.items():
        cb[wl.subpel_idx[c]] = hv.down(g["cost"], np.int32)
    assert n > 100000 and np.array_equal(ca, cb)
    assert ca.max() > 0


def test_step_is_deterministic_across_eager_graph_and_lanes(hv):
    """checksum of checksums over every output buffer: eager serial == HIP-graph replay == 8 fork/join lanes"""
    wl, a, _ = _frames(hv, (1920, 1080), 8, 11)
    a.step()
    hv.sync()
    ref = a.checksum()
    a.step(8)
    hv.sync()
    assert a.checksum() == ref
    g = hv.graph_capture(lambda: a.step(8))
    for _ in range(3):
        hv.graph_launch(g)
    hv.sync()
    assert a.checksum() == ref
    hv.graph_destroy(g)


def test_fused_tu_chain_equals_separate_primitives(hv):
    """272 k transform units at 1080p: tu_forward / tu_reconstruct give the same coefficients, reconstruction and SSD
    as residual -> transform and quantize_inverse -> inverse_transform_add -> ssd"""
    import bench
    from turingcodec_amd.workload import FrameWorkload
    wl = FrameWorkload(1920, 1080, 8, 9)
    a = bench.DeviceFrame(hv, wl, fused_tu=True)
    b = bench.DeviceFrame(hv, wl, fused_tu=False)
    a.step()
    b.step()
    hv.sync()
    for key in a.tu:
        ga, gb = a.tu[key], b.tu[key]
        m = ga["fjobs"].shape[0]
        assert np.array_equal(hv.down(ga["coef"], np.int16), hv.down(gb["coef"], np.int16)), key
        assert np.array_equal(hv.down(ga["rec"], np.uint8), hv.down(gb["rec"], np.uint8)), key
        assert np.array_equal(hv.down(ga["ossd"], np.uint32)[:m], hv.down(gb["ossd"], np.uint32)[:m]), key


def test_tu_chain_roundtrip_property(hv):
    """forward transform -> quantise -> de-quantise -> inverse transform + add reconstructs the source block to within
    the quantiser step (QP 32) for every transform unit of the 1080p workload: SSD(source, recon) stays small and is
    exactly 0 where every level is 0 and the prediction equals the source"""
    wl, a, _ = _frames(hv, (1920, 1080), 8, 3)
    a.step()
    hv.sync()
    for (log2, tr), g in a.tu.items():
        ssd = hv.down(g["ossd"], np.uint32).astype(np.float64)
        n = 1 << log2
        assert np.isfinite(ssd).all() and ssd.mean() / (n * n) < 200.0, (log2, tr, ssd.mean() / (n * n))


def test_bench_multi_rank_path_rehearsal():
    """bench.py's N>1 path (process group, one picture per rank, staged + overlapped reference exchange, barrier,
    max over ranks) with two ranks sharing the one GPU of the test box; gloo stands in for RCCL, which refuses two
    ranks on one device.  The per-rank results must be those of the single-rank run of the same seed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HAVOC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--kernel-reps", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["scaling"] == "weak" and r["value"] > 0
    assert "frame-parallel x2" in r["config"]["parallelism"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--kernel-reps", "1",
                          "--no-cpu-baseline", "--exchange"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert r1["checksum"] == r["checksum"]     # rank 0 encodes the same picture (seed + rank) in both runs


def test_full_size_results_equal_the_reference_library(hv):
    """every 4th job of the 1080p frame through the reference's own havoc functions (oracle/_ref, built from the
    reference sources by oracle/Makefile; x86 JIT tables) on the host, against the GPU results of the same jobs:
    SAD4, SAD, PU SATD, intra predictions, 35-mode intra costs and forward-transform coefficients, ~20 M values"""
    import argparse
    import os
    import bench
    from turingcodec_amd.workload import FrameWorkload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libhavoc_ref.so")):
        pytest.skip("oracle/_ref not built (needs the reference sources at build time)")
    wl = FrameWorkload(1920, 1080, 8, 11)
    dev = bench.DeviceFrame(hv, wl)
    dev.step()
    hv.sync()
    r = bench.cpu_baseline(argparse.Namespace(res="1920x1080", bit_depth=8, seed=11), dev)
    assert r is not None and r["kind"] == "reference"
    p = r["parity_vs_reference"]
    assert "error" not in p, p
    assert p["compared"] > 10_000_000 and p["mismatches"] == 0, p


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line carrying every field of the driver's contract (plus roofline and
    cpu_baseline), with the values the contract fixes"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--tune", "4", "--kernel-reps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["dtype"] == "u8" and r["data"] == "synthetic" and r["unit"] == "frames/s"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) / r["value"] < 0.01
    rf = r["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = r["cpu_baseline"]
    if cb is not None:   # None only when oracle/_ref was never built
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "reference" and cb["cores"] >= 1
        assert cb["parity_vs_reference"]["mismatches"] == 0
