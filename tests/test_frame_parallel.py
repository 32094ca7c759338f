"""CPU tests of the N>1 path: GOP plan and the reference-picture exchange over gloo (world_size 2)."""
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


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_lockstep_plan_is_schedulable(world):
    pics = fp.coding_order(8)
    ready = fp.dependency_ready_step(pics, world)
    # a picture never has to wait more than a few steps for its references in steady state
    late = [r - p.index // world for p, r in zip(pics, ready)]
    assert max(late) <= 1 + 8 // world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _recon_for(pic, n):
    """deterministic stand-in for a picture's reconstruction (what the hot path writes)"""
    rng = np.random.default_rng(1000 + pic.poc)
    return torch.from_numpy(rng.integers(0, 256, n).astype(np.uint8))


def _worker(rank, world, port, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nl, nc = 4096, 1024
    luma, chroma = torch.zeros(nl, dtype=torch.uint8), torch.zeros(nc, dtype=torch.uint8)
    ex = fp.ReferenceExchange(dist, rank, world, luma, chroma, slots=6, n_sops=4)
    for s in range(steps):
        pic = ex.picture_of(s, rank)
        luma.copy_(_recon_for(pic, nl))            # "encode" my picture of this step
        chroma.copy_(_recon_for(pic, nl)[:nc])
        ex.exchange(s)
    out[rank] = (torch.stack(ex.dpb_luma).numpy().copy(), torch.stack(ex.dpb_chroma).numpy().copy(), ex.sent_bytes)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, steps):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    return dict(out)


def test_reference_exchange_gloo_world2_matches_single_process():
    steps2, steps1 = 8, 16            # the same 16 pictures: 2 ranks x 8 steps vs 1 rank x 16 steps
    two = _run(2, steps2)
    one = _run(1, steps1)
    # every rank's DPB mirror is identical, and equals the single-process store
    assert np.array_equal(two[0][0], two[1][0]) and np.array_equal(two[0][1], two[1][1])
    assert np.array_equal(two[0][0], one[0][0]) and np.array_equal(two[0][1], one[0][1])
    # the store holds real reconstructions: slot of POC 8 ((8//2) % 6 = 4)
    pics = fp.coding_order(4)
    p8 = next(p for p in pics if p.poc == 8)
    assert np.array_equal(two[0][0][4], _recon_for(p8, 4096).numpy())
    # only reference pictures were sent (coding order 0 8 4 2 1 3 6 5 7 8+8 ...: count refs among the first 16)
    nref = sum(p.is_reference for p in pics[:16])
    assert two[0][2] + two[1][2] == nref * (4096 + 1024) * 1
