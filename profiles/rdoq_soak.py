"""Soak test of havoc_mi355x_rdoq against the CPU checker: random block counts, sizes, bit depths, state snapshots.
(test infrastructure; run on the GPU box:  python profiles/rdoq_soak.py [rounds])"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rdoq_tools as rt           # noqa: E402
import reflibs                    # noqa: E402
from turingcodec_amd import havoc  # noqa: E402
from turingcodec_amd.havoc import Havoc  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hv = Havoc(stream="new")
orc = reflibs.Oracle()
rng = np.random.default_rng(2026)
blocks_total = bad = 0
for r in range(rounds):
    log2 = int(rng.integers(2, 6))
    bd = int(rng.choice([8, 10, 12]))
    count = int(rng.integers(1, 6000 >> (2 * (log2 - 2)) if log2 > 2 else 6000)) if r % 5 else int(rng.choice([1, 63, 64, 65, 127, 129]))
    src, states, blocks = rt.make_blocks(5000 + r, log2, bd, count, n_states=int(rng.integers(1, 20)))
    want, want_cbf = rt.run_cpu(orc, src, states, blocks)
    got, got_cbf = hv.rdoq(bd, log2, src, states, rt.device_jobs(blocks, havoc.rdoq_lambda))
    ok = np.array_equal(got, want) and np.array_equal(got_cbf, want_cbf)
    blocks_total += count
    bad += 0 if ok else 1
    print(f"round {r}: log2 {log2} bd {bd} blocks {count}: {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{blocks_total} blocks in {rounds} rounds, {bad} mismatching rounds")
sys.exit(1 if bad else 0)
