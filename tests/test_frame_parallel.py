"""CPU tests of the N>1 path: the GOP plan, the dependency-aware schedule and the reference-picture exchange over gloo
(world_size 2 against world_size 1)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from turingcodec_amd import frame_parallel as fp


def test_coding_order_matches_reference_sop():
    pics = fp.coding_order(2)
    assert [p.poc for p in pics[:9]] == [0, 8, 4, 2, 1, 3, 6, 5, 7]       # turing/InputQueue.cpp:370-379
    assert [p.is_reference for p in pics[:9]] == [True, True, True, True, False, False, True, False, False]
    by = {p.poc: p for p in pics}
    assert by[8].refs == (0,) and by[4].refs == (0, 8) and by[2].refs == (0, 4, 8)
    assert by[1].refs == (0, 2, 4, 8) and by[6].refs == (0, 4, 8) and by[7].refs == (0, 6, 8)
    assert by[16].refs == (8,) and by[12].refs == (8, 16)
    # every referenced picture is a reference picture and precedes its users in coding order
    for p in pics:
        for q in p.refs:
            assert by[q].is_reference and by[q].index < p.index
    # searched lists: nearest past picture / nearest future picture (anchors have no future one)
    assert (by[4].l0, by[4].l1) == (0, 8) and (by[3].l0, by[3].l1) == (2, 4) and (by[8].l0, by[8].l1) == (0, 0)


@pytest.mark.parametrize("world,lag", [(1, 1), (2, 1), (3, 1), (4, 1), (8, 1), (8, 2), (4, 2)])
def test_dag_schedule_respects_dependencies(world, lag):
    n_sops = 12
    s = fp.DagSchedule(world, n_sops=n_sops, lag=lag)
    assert s.lag == lag
    nslots = s.slots_for_sequence()
    seen = {}
    live = {}       # DPB slot -> poc currently held
    for t in range(nslots):
        row = s.slot(t)
        assert len(row) == world
        for p in row:
            if p is None:
                continue
            assert p.poc not in seen
            for q in p.refs:   # the wait-for-reference rule: every reference finished in an EARLIER slot ...
                assert q in seen and seen[q] < t, (p.poc, q)
                # ... and `lag` slots earlier when it comes from another rank: its broadcast has a whole slot to land
                if s.rank_of[q] != row.index(p):
                    assert seen[q] + lag <= t, (p.poc, q, seen[q], t)
            seen[p.poc] = t
            assert s.rank_of[p.poc] == row.index(p)
            if world > 1 and p.poc % 8 == 0:
                assert row.index(p) == 0      # the anchor chain stays on rank 0
        for p in row:
            if p is not None and p.is_reference:
                d = s.dpb_slot[p.poc]
                if d in live:   # a mirror slot is re-used only when every user of its previous picture ran before slot t
                    old = live[d]
                    users = [u for u in s.pics if old in u.refs]
                    assert all(seen.get(u.poc, 1 << 30) < t for u in users), (p.poc, old)
                live[d] = p.poc
    assert len(seen) == 1 + 8 * n_sops
    # a picture's references are still in the mirror when it runs: the slot was not handed to a later picture before
    by_slot_owner = {}
    for t in range(nslots):
        for p in s.slot(t):
            if p is not None and p.is_reference:
                by_slot_owner.setdefault(s.dpb_slot[p.poc], []).append((t, p.poc))
    for p in s.pics:
        for q in p.refs:
            hist = by_slot_owner[s.dpb_slot[q]]
            holder = max((tt, poc) for tt, poc in hist if tt < seen[p.poc])
            assert holder[1] == q, (p.poc, q, holder)


def test_eight_ranks_reach_the_level_skewed_pipeline():
    s = fp.DagSchedule(8, lag=1)
    # steady state: per slot POCs 1,3,5,7 of SOP k, 2,6 of SOP k+1, 4 of SOP k+2 and the anchor of SOP k+3
    for t in range(6, 40):
        pocs = sorted(p.poc for p in s.slot(t))
        k = pocs[0] // 8
        assert pocs == [8 * k + 1, 8 * k + 3, 8 * k + 5, 8 * k + 7, 8 * k + 10, 8 * k + 14, 8 * k + 20, 8 * k + 32]
    # the default for 8 ranks on an endless sequence gives every broadcast a slot to land (lag 2): twice the skew between
    # the levels, still one SOP = 8 pictures per slot, within the default window and mirror
    s = fp.DagSchedule(8)
    assert s.lag == 2
    for t in range(14, 60):
        pocs = sorted(p.poc for p in s.slot(t))
        k = pocs[0] // 8
        assert pocs == [8 * k + 1, 8 * k + 3, 8 * k + 5, 8 * k + 7, 8 * k + 18, 8 * k + 22, 8 * k + 36, 8 * k + 56], pocs
    for world in (1, 2, 4, 8):   # no idle rank in steady state at any width
        s = fp.DagSchedule(world)
        assert all(p is not None for t in range(20, 90) for p in s.slot(t))
    assert fp.DagSchedule(8, n_sops=4).slots_for_sequence() == 8      # 33 pictures: fill + drain dominate
    assert fp.DagSchedule(1, n_sops=4).slots_for_sequence() == 33


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


NL, NC = 4096, 1024


def _source(poc):
    rng = np.random.default_rng(1000 + poc)
    return torch.from_numpy(rng.integers(0, 256, NL + 2 * NC).astype(np.uint8))


def _encode(poc, ref0, ref1):
    """deterministic stand-in for the hot path: the reconstruction is a function of the source and of BOTH references'
    reconstructions, so a picture that started before a reference arrived produces a different result"""
    out = _source(poc).to(torch.int32)
    if ref0 is not None:
        out = out + 3 * ref0.to(torch.int32) + 5 * ref1.to(torch.int32).roll(1)
    return (out % 251).to(torch.uint8)


def _expected(n_sops):
    """per-POC reconstructions computed sequentially in coding order (no scheduling involved)"""
    rec = {}
    for p in fp.coding_order(n_sops):
        rec[p.poc] = _encode(p.poc, rec[p.l0] if p.refs else None, rec[p.l1] if p.refs else None)
    return rec


def _worker(rank, world, port, n_sops, out, lag=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sched = fp.DagSchedule(world, n_sops=n_sops, lag=lag)
    ex = fp.ReferenceExchange(dist, rank, sched, NL, NC, torch.zeros(1, dtype=torch.uint8))
    want = _expected(n_sops)
    sums = {}
    early = []
    t = 0
    while not sched.finished(t):
        pic = ex.picture_of(t)
        rec = None
        if pic is not None:
            if pic.refs:
                s0, s1 = ex.refs(pic)
                r0, r1 = ex.dpb[s0].clone(), ex.dpb[s1].clone()      # read from the MIRROR, whoever encoded them
                if not (torch.equal(r0, want[pic.l0]) and torch.equal(r1, want[pic.l1])):
                    early.append(pic.poc)                             # a reference had not arrived (or was overwritten)
                rec = _encode(pic.poc, r0, r1)
            else:
                rec = _encode(pic.poc, None, None)
            sums[pic.poc] = int(rec.to(torch.int64).sum()) * 1000003 + int(rec[::7].to(torch.int64).sum())
            ex.stage(t, (rec[:NL], rec[NL:NL + NC], rec[NL + NC:]))
        ex.send(t)
        t += 1
    out[rank] = (sums, early, ex.sent_bytes, ex.broadcasts, t)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_sops, lag=None):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_sops, out, lag), nprocs=world, join=True)
    return dict(out)


def test_frame_parallel_gloo_world2_equals_single_rank_per_poc():
    n_sops = 3
    two, one = _run(2, n_sops), _run(1, n_sops)
    want = {poc: int(r.to(torch.int64).sum()) * 1000003 + int(r[::7].to(torch.int64).sum()) for poc, r in _expected(n_sops).items()}
    # no picture started before its references had arrived in the local mirror
    assert two[0][1] == [] and two[1][1] == [] and one[0][1] == []
    # every picture was encoded exactly once across the ranks, and its result does not depend on the number of ranks
    merged = dict(two[0][0])
    assert not (set(merged) & set(two[1][0]))
    merged.update(two[1][0])
    assert merged == one[0][0] == want
    # only reference pictures were broadcast: the IDR + 4 per SOP, each once
    nref = 1 + 4 * n_sops
    assert two[0][3] == two[1][3] == nref
    assert two[0][2] + two[1][2] == nref * (NL + 2 * NC)
    assert two[0][4] == fp.DagSchedule(2, n_sops=n_sops).slots_for_sequence() and one[0][4] == 1 + 8 * n_sops


def test_frame_parallel_gloo_world4_lag2_equals_sequential_coding():
    """the deeper schedule (every reference from another rank two slots ahead of its users, the anchor chain local to rank 0, 40 mirror
    slots) over gloo with four ranks: every picture's reconstruction equals the one sequential coding gives"""
    n_sops = 5
    res = _run(4, n_sops, lag=2)
    want = {poc: int(r.to(torch.int64).sum()) * 1000003 + int(r[::7].to(torch.int64).sum()) for poc, r in _expected(n_sops).items()}
    merged = {}
    for r in range(4):
        assert res[r][1] == [], (r, res[r][1])
        assert not (set(merged) & set(res[r][0]))
        merged.update(res[r][0])
    assert merged == want
    assert all(poc % 8 == 0 for poc in res[0][0] if poc % 8 == 0) and set(p for p in want if p % 8 == 0) <= set(res[0][0])      # anchors on rank 0
    assert res[0][4] == fp.DagSchedule(4, n_sops=n_sops, lag=2).slots_for_sequence()


# ---- the exchange in CTU-row bands --------------------------------------------------------------------------------------------
BH, BPAD, BLS, BCS = 320, 32, 128, 64          # 5 CTU rows; padded planes 384 x 128 (luma), 192 x 64 (chroma)
BNL, BNC = (BH + 2 * BPAD) * BLS, (BH // 2 + BPAD) * BCS


def test_band_plan_tiles_the_padded_planes_and_follows_the_reference_rule():
    for height, pad, rows in ((1080, 96, 4), (2160, 96, 4), (320, 32, 2), (64, 16, 4), (4320, 96, 8)):
        plan = fp.BandPlan(height, pad, 4096, 2048, band_ctu_rows=rows)
        nl, nc = plan.luma_rows_total * plan.ls, plan.chroma_rows_total * plan.cs
        covered = np.zeros(nl + 2 * nc, np.int32)
        for b in range(plan.n_bands):
            for lo, hi in plan.pieces(b, nl, nc):
                covered[lo:hi] += 1
        assert (covered == 1).all()                                   # every sample of the three planes in exactly one band
        assert plan.luma_rows(0)[0] == 0 and plan.luma_rows(plan.n_bands - 1)[1] == height + 2 * pad
        for r in range(plan.ctu_rows):                                # turing/TaskEncodeSubstream.cpp:71-95: the reference must be there 3 CTU rows below
            need = plan.bands_needed(r)
            assert plan.band_of_ctu_row(r + 3) == need - 1
            assert plan.rows_ready(need) > r and (need == 1 or plan.rows_ready(need - 1) <= r)
        assert plan.rows_ready(plan.n_bands) == plan.ctu_rows and plan.rows_ready(0) == 0


def _worker_bands(rank, world, port, n_sops, out, async_op):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sched = fp.DagSchedule(world, n_sops=n_sops)
    ex = fp.ReferenceExchange(dist, rank, sched, BNL, BNC, torch.zeros(1, dtype=torch.uint8))
    plan = fp.BandPlan(BH, BPAD, BLS, BCS, band_ctu_rows=2)
    ex.set_bands(plan)
    want = _expected_sized(n_sops)
    sums, early, partial_ok = {}, [], True
    t = 0
    while not sched.finished(t):
        pic = ex.picture_of(t)
        rec = None
        if pic is not None:
            if pic.refs:
                s0, s1 = ex.refs(pic)
                # band mode: every CTU row of this picture may start -- all bands of both references have arrived
                if ex.rows_ready(pic.l0) != plan.ctu_rows or ex.rows_ready(pic.l1) != plan.ctu_rows:
                    early.append(pic.poc)
                r0, r1 = ex.dpb[s0].clone(), ex.dpb[s1].clone()
                if not (torch.equal(r0, want[pic.l0]) and torch.equal(r1, want[pic.l1])):
                    early.append(pic.poc)
                rec = _encode_sized(pic.poc, r0, r1)
            else:
                rec = _encode_sized(pic.poc, None, None)
            sums[pic.poc] = int(rec.to(torch.int64).sum()) * 1000003 + int(rec[::7].to(torch.int64).sum())
        for b in range(plan.n_bands):      # a band leaves as soon as it is "deblocked and padded"; the later bands are still being computed
            if rec is not None:
                ex.stage_band(t, b, (rec[:BNL], rec[BNL:BNL + BNC], rec[BNL + BNC:]))
            works = ex.send_band(t, b, async_op=async_op)
            for w in works:
                w.wait()
            # after band b: the mirrors of this slot's reference pictures hold bands 0 .. b of the final reconstruction, and say so
            for src in range(world):
                q = ex.picture_of(t, src)
                if q is None or not q.is_reference:
                    continue
                buf = ex.dpb[ex.slot_of(q.poc)]
                for bb in range(b + 1):
                    for lo, hi in plan.pieces(bb, BNL, BNC):
                        partial_ok &= bool(torch.equal(buf[lo:hi], want[q.poc][lo:hi]))
                partial_ok &= ex.rows_ready(q.poc) == plan.rows_ready(b + 1)
        t += 1
    out[rank] = (sums, early, ex.sent_bytes, ex.broadcasts, t, partial_ok)
    dist.barrier()
    dist.destroy_process_group()


def _source_sized(poc):
    rng = np.random.default_rng(2000 + poc)
    return torch.from_numpy(rng.integers(0, 256, BNL + 2 * BNC).astype(np.uint8))


def _encode_sized(poc, ref0, ref1):
    out = _source_sized(poc).to(torch.int32)
    if ref0 is not None:
        out = out + 3 * ref0.to(torch.int32) + 5 * ref1.to(torch.int32).roll(1)
    return (out % 251).to(torch.uint8)


def _expected_sized(n_sops):
    rec = {}
    for p in fp.coding_order(n_sops):
        rec[p.poc] = _encode_sized(p.poc, rec[p.l0] if p.refs else None, rec[p.l1] if p.refs else None)
    return rec


def _run_bands(world, n_sops, async_op):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_bands, args=(world, _free_port(), n_sops, out, async_op), nprocs=world, join=True)
    return dict(out)


@pytest.mark.parametrize("async_op", [False, True])
def test_band_exchange_gloo_world2_equals_single_rank_per_poc(async_op):
    """the reference pictures leave in CTU-row bands (3 broadcasts per band: the band's rows of Y, Cb, Cr): per-POC checksums equal to one rank's and to sequential
    coding, the mirrors hold exactly the bands sent so far, and rows_ready() follows the reference's 3-rows-below rule"""
    n_sops = 2
    two, one = _run_bands(2, n_sops, async_op), _run_bands(1, n_sops, async_op)
    want = {poc: int(r.to(torch.int64).sum()) * 1000003 + int(r[::7].to(torch.int64).sum()) for poc, r in _expected_sized(n_sops).items()}
    assert two[0][1] == [] and two[1][1] == [] and one[0][1] == []
    assert two[0][5] and two[1][5] and one[0][5]
    merged = dict(two[0][0])
    assert not (set(merged) & set(two[1][0]))
    merged.update(two[1][0])
    assert merged == one[0][0] == want
    nref, bands = 1 + 4 * n_sops, fp.BandPlan(BH, BPAD, BLS, BCS, band_ctu_rows=2).n_bands
    assert two[0][3] == two[1][3] == nref * bands * 3              # every band of every reference picture once, three planes each
    assert two[0][2] + two[1][2] == nref * (BNL + 2 * BNC)         # and the bands add up to the whole padded picture


def test_bench_refuses_a_world_that_is_not_what_gpus_asked_for():
    """`--gpus N` is checked against the ranks the launcher started: a 1-rank run of `--gpus 2` must not print a line"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 2 and "--gpus 2" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_decision_pipeline_two_rank_rehearsal_equals_one_rank_with_and_without_bands():
    """bench.py --decisions 3: the frame-parallel pipeline driving the DECISION step (a DecisionPicture.step per picture, references read from the DPB mirror).
    One rank with the RCCL exchange (whole pictures, then CTU-row bands) against two ranks sharing the one GPU over gloo: per-POC checksums of the reconstructions
    (all three planes, borders included) equal -- a picture's result does not depend on who encodes it or on how its references travelled."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")

    def run(gpus, extra, env=None):
        out = subprocess.run([sys.executable, bench, "--decisions", "3", "--gpus", str(gpus), "--res", "416x240", "--pictures", "17"] + extra, capture_output=True, text=True,
                             env=dict(os.environ, **(env or {})), timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(1, ["--exchange"])
    one_bands = run(1, ["--exchange", "--bands", "2"])
    two = run(2, ["--bands", "2"], {"HAVOC_BENCH_BACKEND": "gloo"})
    assert one["pictures"] == one_bands["pictures"] == two["pictures"] == 17 and len(one["poc_checksums"]) == 17
    assert one["poc_checksums"] == one_bands["poc_checksums"] == two["poc_checksums"]
    assert len(set(one["poc_checksums"].values())) == 17                       # every picture of the sequence is its own
    assert one_bands["config"]["broadcasts"] == 9 * 2 * 3 and one["config"]["broadcasts"] == 9
    assert two["slots"] < one["slots"]                                          # two ranks: fewer time slots for the same sequence


def test_the_band_plan_never_lets_a_row_start_that_the_device_gate_would_hold():
    """host rule against device rule: BandPlan.rows_ready (the reference's three CTU rows below, in bands) and havoc_mi355x_search_gate's condition for CTU row r --
    rows_ready >= min((r + 2) * 64, H + 72), with the counter a banded producer publishes after band b (its last row - 4, once more - 4 for the rows whose fractional
    planes have their filter taps): every row the plan lets start, the gate lets start"""
    from turingcodec_amd.frame_parallel import BandPlan
    for H in (240, 480, 1080, 2160):
        for rows in (1, 2, 3, 4):
            plan = BandPlan(H, 96, 4096, 2048, band_ctu_rows=rows)
            for arrived in range(plan.n_bands + 1):
                if arrived == 0:
                    gate = 0
                elif arrived == plan.n_bands:
                    gate = H + 96 - 4
                else:
                    gate = min(H, arrived * rows * 64) - 8
                for r in range(plan.rows_ready(arrived)):
                    assert gate >= min((r + 2) * 64, H + 72), (H, rows, arrived, r)
            assert plan.rows_ready(plan.n_bands) == plan.ctu_rows


@pytest.mark.gpu
def test_one_sequence_with_its_dependencies_on_virtual_ranks_equals_the_one_rank_pipeline():
    """bench.py --decisions 4: the frame-parallel schedule of K ranks executed by K host threads / contexts on ONE GPU, sharing one DPB mirror (round 5): per-POC
    checksums equal to the one-rank pipeline of --decisions 3, for 2 and 8 virtual ranks -- pictures really predict from the pictures other contexts reconstructed,
    and nothing depends on who ran what when"""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")

    def run(extra):
        out = subprocess.run([sys.executable, bench, "--res", "416x240", "--pictures", "17", "--poc-checksums"] + extra, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    one = run(["--decisions", "3", "--gpus", "1"])
    two = run(["--decisions", "4", "--virtual-ranks", "2"])
    eight = run(["--decisions", "4", "--virtual-ranks", "8"])
    assert one["pictures"] == two["pictures"] == eight["pictures"] == 17 and len(two["poc_checksums"]) == 17
    assert one["poc_checksums"] == two["poc_checksums"] == eight["poc_checksums"]
    assert eight["slots"] < two["slots"] < one["slots"]
    # ... and with the reconstructions entering the mirror BAND BY BAND while their lower rows are still searched, the pictures predicting from them following them down
    # the picture (DecisionPicture.step_banded + havoc_mi355x_search_gate): the same pictures, sample for sample
    # (queued by a thread per context, or the whole sequence by ONE thread -- nothing on the host then waits for the device before the end)
    # (at most four contexts here: three streams each stay below the 24 hardware queues the pipeline asks HIP for, so no stream that waits shares a queue with anything)
    for ranks, rows, issue in ((2, 1, "threads"), (4, 2, "single"), (4, 1, "threads"), (3, 1, "single")):
        banded = run(["--decisions", "4", "--virtual-ranks", str(ranks), "--vr-bands", str(rows), "--vr-issue", issue])
        assert banded["poc_checksums"] == one["poc_checksums"], (ranks, rows, issue)
