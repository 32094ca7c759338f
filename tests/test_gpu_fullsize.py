"""Full-size GPU checks through size-independent properties (the oracle would take minutes at these sizes):
the bench workload at BASELINE.json's resolutions must give identical results on independent code paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd import Havoc
    return Havoc(0, stream="new")


def _frames(hv, res, bit_depth, seed, qp=32):
    import bench
    from turingcodec_amd import step
    from turingcodec_amd.workload import FrameWorkload
    w, h = res
    wl = FrameWorkload(w, h, bit_depth, seed, qp=qp)
    return wl, step.DeviceFrame(hv, wl, use_planes=True), step.DeviceFrame(hv, wl, use_planes=False)


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
    from turingcodec_amd import step
    from turingcodec_amd.workload import FrameWorkload
    wl = FrameWorkload(1920, 1080, 8, 9)
    a = step.DeviceFrame(hv, wl, fused_tu=True)
    b = step.DeviceFrame(hv, wl, fused_tu=False)
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


def _bench_json(cmd, env, root, timeout=900):
    import json
    import subprocess
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_frame_parallel_rehearsal_equals_single_rank_per_poc():
    """bench.py's N>1 path (process group, dependency-aware schedule, references read from the DPB mirror, staged +
    overlapped broadcasts, barrier, max over ranks) with two ranks sharing the one GPU of the test box; gloo stands in
    for RCCL, which refuses two ranks on one device.  A fixed 17-picture sequence (IDR + 2 SOPs): the reconstruction
    checksum of every POC must be the one the single-rank run produces -- i.e. no picture started before its references
    had arrived, and who encodes a picture does not matter."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HAVOC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--kernel-reps", "1", "--tune", "0", "--scaling", "strong", "--pictures", "17", "--poc-checksums", "--no-cpu-baseline",
              "--res", "640x360"]
    r2 = _bench_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29577", os.path.join(root, "bench.py"), "--gpus", "2"] + common, env, root)
    assert r2["n_gpus"] == 2 and r2["scaling"] == "strong" and r2["value"] > 0
    assert "frame-parallel x2" in r2["config"]["parallelism"]
    r1 = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--exchange"] + common, env, root)
    assert r1["n_gpus"] == 1 and len(r1["poc_checksums"]) == 17
    # every picture of the 2-rank run (the ranks' checksums are gathered) must agree with the single-rank run
    assert len(r2["poc_checksums"]) == 17
    for poc, c in r2["poc_checksums"].items():
        assert r1["poc_checksums"][poc] == c, poc
    # the dependency is real: pictures of one hierarchy level have different reconstructions
    assert len(set(r1["poc_checksums"].values())) >= 5
    from turingcodec_amd.frame_parallel import DagSchedule
    assert r2["steps"] == DagSchedule(2, n_sops=2).slots_for_sequence() and r1["steps"] == 17


def test_bench_frame_parallel_rehearsal_eight_ranks_lag_two():
    """the 8-rank schedule the driver's scaling run uses (anchor chain on rank 0, every other reference two slots ahead of its
    users, 40 mirror slots): eight gloo ranks on the one GPU, a 49-picture sequence, every POC's reconstruction equal to the
    single-rank run's"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HAVOC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--kernel-reps", "1", "--tune", "0", "--scaling", "strong", "--pictures", "49", "--poc-checksums", "--no-cpu-baseline",
              "--res", "416x240"]
    r8 = _bench_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                      "--master-port", "29579", os.path.join(root, "bench.py"), "--gpus", "8", "--lag", "2"] + common, env, root, timeout=1500)
    assert r8["n_gpus"] == 8 and len(r8["poc_checksums"]) == 49
    r1 = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--exchange"] + common, env, root)
    assert r1["poc_checksums"] == r8["poc_checksums"]
    from turingcodec_amd.frame_parallel import DagSchedule
    assert r8["steps"] == DagSchedule(8, n_sops=6, lag=2).slots_for_sequence()


def test_bench_weak_scaling_rehearsal_two_ranks():
    """the driver's N>1 invocation (weak scaling: steady state, one picture per rank per slot) with two gloo ranks on one GPU"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HAVOC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _bench_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", "29578", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "6", "--kernel-reps", "1",
                     "--tune", "0", "--min-seconds", "0.05", "--res", "640x360"], env, root)
    assert r["n_gpus"] == 2 and r["steps"] == 6 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["pictures_per_timed_block"] == 12.0     # both ranks busy in every timed slot


def test_bench_weak_scaling_rehearsal_eight_ranks():
    """eight gloo ranks on one GPU, the default (lag 2) schedule: the untimed pipeline fill puts every timed slot in the steady state --
    8 pictures per slot although the driver's warm-up is shorter than the fill"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HAVOC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _bench_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                     "--master-port", "29580", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--kernel-reps", "1",
                     "--tune", "0", "--min-seconds", "0.0", "--res", "416x240"], env, root, timeout=1500)
    assert r["n_gpus"] == 8 and r["steps"] == 4 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["pictures_per_timed_block"] == 32.0
    assert "lag 2" in r["config"]["parallelism"]


def _parity_vs_reference(hv, res, bit_depth, qp, mix="ra", seed=11, min_values=10_000_000):
    import argparse
    import os
    import bench
    from turingcodec_amd import step
    from turingcodec_amd.workload import FrameWorkload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "libhavoc_ref.so")):
        pytest.skip("oracle/_ref not built (needs the reference sources at build time)")
    wl = FrameWorkload(res[0], res[1], bit_depth, seed, qp=qp, mix=mix)
    dev = step.DeviceFrame(hv, wl)
    dev.step()
    hv.sync()
    r = bench.cpu_baseline(argparse.Namespace(res=f"{res[0]}x{res[1]}", bit_depth=bit_depth, seed=seed, qp=qp, mix=mix, rdoq=1), dev)
    assert r is not None and r["kind"] == "reference"
    p = r["parity_vs_reference"]
    assert "error" not in p, p
    assert p["compared"] > min_values and p["mismatches"] == 0, p
    return p


def test_full_size_results_equal_the_reference_library(hv):
    """every 4th job of the 1080p frame (BASELINE.json configs[1]: 8-bit QP32) through the reference's own havoc functions
    (oracle/_ref, built from the reference sources by oracle/Makefile; x86 JIT tables) on the host, against the GPU results
    of the same jobs: SAD4, SAD, PU SATD, uni / bi interpolations (luma + chroma), SubtractBi, sub-pel candidate costs, intra
    predictions, 35-mode intra costs, forward coefficients, the levels and coded-block flags of Rdoq::runQuantisation (the
    reference's own Rdoq.cpp), reconstructions and their SSDs"""
    p = _parity_vs_reference(hv, (1920, 1080), 8, 32)
    for name in ("pred_bi8", "pred_bi4", "subtract_bi", "rec_3_0", "ssd_3_0", "level_2_1", "cbf_5_0", "cbf_2_1"):
        assert name in p["groups"], name


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_4k_qp27_results_equal_the_reference_library(hv, bit_depth):
    """BASELINE.json configs[2] and [3]: 3840x2160 random-access QP27, 8-bit and 10-bit (16-bit sample kernels)"""
    _parity_vs_reference(hv, (3840, 2160), bit_depth, 27, min_values=40_000_000)


def test_8k_qp32_results_equal_the_reference_library(hv):
    """BASELINE.json configs[4]: 7680x4320 8-bit random-access QP32 (the resolution of the 8-GPU configuration) on one GPU against the
    reference's compiled sources, the same comparison as at 1080p and 4K"""
    _parity_vs_reference(hv, (7680, 4320), 8, 32, seed=19, min_values=40_000_000)


def test_all_intra_fast_mix_equals_the_reference_library(hv):
    """BASELINE.json configs[0]: 640x360 all-intra QP32 speed=fast -- intra + TU chain with havoc_quantize IN the timed
    chain (no RDOQ at fast, turing/Reconstruct.cpp:310-311)"""
    p = _parity_vs_reference(hv, (640, 360), 8, 32, mix="ai", seed=7, min_values=1_000_000)
    assert "level_3_0" in p["groups"] and "sad4" not in p["groups"]


def test_8k_frame_properties(hv):
    """BASELINE.json configs[4] resolution (7680x4320 8-bit QP32) on one GPU, through size-independent properties: the two
    independent sub-pel routes agree for every candidate, the fused TU chain's reconstruction error stays within the
    quantiser step, the final reconstruction pass covers every sample of the picture exactly once and serial == 8-lane
    execution"""
    wl, a, b = _frames(hv, (7680, 4320), 8, 19)
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
    assert n > 2_000_000 and np.array_equal(ca, cb)
    for (log2, tr), g in a.tu.items():
        ssd = hv.down(g["ossd"], np.uint32).astype(np.float64)
        assert ssd.mean() / (1 << (2 * log2)) < 200.0, (log2, tr)
    pl, st = wl.plane_len, wl.stride
    rec = hv.down(a.luma[3 * pl:4 * pl], np.uint8).reshape(-1, st)[96:96 + 4320, 96:96 + 7680]
    src = wl.luma[:pl].reshape(-1, st)[96:96 + 4320, 96:96 + 7680]
    assert np.abs(rec.astype(np.int32) - src).mean() < 40 and (rec != 0).mean() > 0.95   # every block was reconstructed
    ref = a.checksum()
    a.step(8)
    hv.sync()
    assert a.checksum() == ref


def test_scan_inside_tu_forward_equals_the_separate_scan(hv):
    """havoc_mi355x_tu_forward_scan + havoc_mi355x_rdoq_prescanned against tu_forward + havoc_mi355x_rdoq on the 1080p picture's 16x16 and 32x32
    transform blocks: coefficients, levels, coded-block flags, reconstructions and SSDs identical (8- and 10-bit)"""
    import bench
    from turingcodec_amd import step
    from turingcodec_amd.workload import FrameWorkload
    for bd, qp in ((8, 32), (10, 27)):
        wl = FrameWorkload(1920, 1080, bd, 11, qp=qp)
        a = step.DeviceFrame(hv, wl, scan_in_forward=True)
        b = step.DeviceFrame(hv, wl, scan_in_forward=False)
        a.step()
        b.step()
        hv.sync()
        assert len(a.launches) == len(b.launches)
        for key in a.tu:
            for name, dt in (("coef", np.int16), ("level", np.int16), ("cbf", np.int32), ("rec", wl.dtype), ("ossd", np.uint32)):
                assert np.array_equal(hv.down(a.tu[key][name], dt), hv.down(b.tu[key][name], dt)), (bd, key, name)
        assert a.checksum() == b.checksum()


def test_merged_prediction_launches_equal_the_per_class_launches(hv):
    """havoc_mi355x_pred_uni_classes / pred_bi_classes (all four size classes of a job table in one launch) against one launch per width
    class, on the 1080p picture's luma and chroma, uni and bi tables: identical output buffers"""
    import bench
    from turingcodec_amd import step
    from turingcodec_amd.workload import FrameWorkload
    wl = FrameWorkload(1920, 1080, 8, 11)
    a = step.DeviceFrame(hv, wl, pred_launches="merged")
    b = step.DeviceFrame(hv, wl, pred_launches="classes")
    a.step()
    b.step()
    hv.sync()
    assert sum(name.startswith("pred_") for name, _ in a.launches) == 4 < sum(name.startswith("pred_") for name, _ in b.launches)
    for name in ("pred", "cpred", "bi", "cbi", "sbi"):
        assert np.array_equal(hv.down(getattr(a, name), wl.dtype), hv.down(getattr(b, name), wl.dtype)), name
    assert a.checksum() == b.checksum()


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line carrying every field of the driver's contract (plus roofline and
    cpu_baseline), with the values the contract fixes"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # --decisions 0: the decision-driven path (two more processes, ~40 s) has its own tests (test_decisions.py, test_search.py)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--tune", "4", "--kernel-reps", "1", "--decisions", "0"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) < 8000, len(lines[0])      # the driver keeps 8 KB of the line: round 4's 41 KB line went unparsed (VERDICT r4 next #1c)
    r = json.loads(lines[0])
    assert r["parity"] == "green" and os.path.exists(os.path.join(root, r["detail"]))
    detail = json.load(open(os.path.join(root, r["detail"])))
    assert detail["value"] == r["value"] and "note" in detail["roofline"] and "whole_step" in detail
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["dtype"] == "u8" and r["data"] == "synthetic" and r["unit"] == "frames/s"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) / r["value"] < 0.01
    rf = r["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["frac"] <= 1.0      # unique bytes: a fraction above 1 is not a roofline
    if rf.get("valu"):
        assert 0 < rf["valu"]["frac"] <= 1.0 and rf["valu"]["peak"] == 1228.8
    cb = r["cpu_baseline"]
    if cb is not None:   # None only when oracle/_ref was never built
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "reference" and cb["cores"] >= 1
        assert cb["parity_vs_reference"]["mismatches"] == 0


@pytest.mark.gpu
def test_bench_reference_exchange_over_rccl_single_rank():
    """the collective path itself on the GPU box: process group on the `nccl` backend (= RCCL), staging, DPB mirror and the broadcasts of the
    reference pictures, with the one rank a 1-GPU box has.  Per-POC reconstructions must equal the run without a process
    group, and reference pictures must really have gone through the broadcast call."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HAVOC_BENCH_BACKEND", None)
    common = ["--kernel-reps", "1", "--tune", "0", "--scaling", "strong", "--pictures", "17", "--poc-checksums", "--no-cpu-baseline", "--res", "640x360", "--decisions", "0"]
    rccl = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--exchange"] + common, env, root)
    assert rccl["n_gpus"] == 1 and len(rccl["poc_checksums"]) == 17
    assert "RCCL" in rccl["config"]["parallelism"] and ", 0 reference pictures broadcast" not in rccl["config"]["parallelism"]
    env_gloo = dict(env, HAVOC_BENCH_BACKEND="gloo")
    gloo = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--exchange"] + common, env_gloo, root)
    assert rccl["poc_checksums"] == gloo["poc_checksums"]
