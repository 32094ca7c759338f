"""Generates tests/golden/rdoq_golden.npz: inputs of tests/rdoq_tools.make_blocks and the outputs of the REFERENCE's own
Rdoq::runQuantisation (turing/Rdoq.cpp compiled into oracle/_ref by oracle/Makefile, driven through oracle/ref_shim_rdoq.cpp).
Run in the build container (needs /root/reference at build time of oracle/_ref):  python tests/golden/make_rdoq_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import reflibs       # noqa: E402
import rdoq_tools    # noqa: E402

CASES = [(2, 8, 300), (3, 8, 200), (4, 8, 100), (5, 8, 60), (2, 10, 100), (3, 10, 100), (4, 10, 60), (5, 10, 40)]


def main():
    ref = reflibs.Reference()
    init = [ref.rdoq_initial_states(qp, t) for qp, t in ((22, 0), (32, 1), (37, 2))]
    out = {}
    for log2, bd, count in CASES:
        src, states, blocks = rdoq_tools.make_blocks(1000 + 10 * log2 + bd, log2, bd, count, initial_states=init)
        dst, cbf = rdoq_tools.run_cpu(ref, src, states, blocks)
        k = f"l{log2}b{bd}"
        out[k + ".levels"] = dst
        out[k + ".cbf"] = cbf
        out[k + ".src_crc"] = np.array([np.bitwise_xor.reduce(src.view(np.uint16).astype(np.uint32) * np.arange(1, len(src) + 1, dtype=np.uint32))], np.uint32)
    out["initial_states"] = np.stack(init)
    np.savez_compressed(os.path.join(HERE, "rdoq_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
