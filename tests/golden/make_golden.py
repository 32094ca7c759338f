#!/usr/bin/env python3
"""Generates tests/golden/havoc_golden.npz: the parity suite's inputs, job tables and the outputs of the REFERENCE's
own C functions (havoc C_REF|C_OPT tables, compiled from /root/reference/havoc by oracle/Makefile into
oracle/_ref/libhavoc_ref.so).  Run in the build container, where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

The .npz holds data only (inputs + expected outputs); nothing of the reference's source travels with it.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import reflibs  # noqa: E402
import suite  # noqa: E402

SEED = 20260927


def main():
    assert reflibs.have_reference(), "build oracle/_ref first: make -C oracle ref"
    d = suite.make_inputs(SEED)
    exp = suite.run(suite.LoopImpl(reflibs.Reference(0)), d)
    blob = {"in." + k: v for k, v in d.items()}
    blob.update({"out." + k: v for k, v in exp.items()})
    blob["seed"] = np.array([SEED])
    path = os.path.join(HERE, "havoc_golden.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes;", len(d), "inputs,", len(exp), "outputs")


if __name__ == "__main__":
    main()
