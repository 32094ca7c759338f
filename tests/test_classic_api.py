"""The classic per-block table API (include/havoc/*.h + libhavoc_classic.so) as a drop-in for libhavoc.a.

CPU: (1) every public populate/get/new_code symbol the reference library defines is defined by ours with the
identical mangled name; (2) a client written against the REFERENCE's headers (oracle/ref_shim.cpp: it populates all
24 tables the way turing/StateFunctionTables.h:63-91 does) compiles unchanged against OUR headers and links.
GPU: that client, driven through the parity suite, reproduces the golden vectors bit-exactly (every table entry runs
one-job launches of the batch kernels)."""
import os
import subprocess

import numpy as np
import pytest

import suite
from reflibs import REF_SO, Reference
from test_golden import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "turingcodec_amd")
CLIENT = os.path.join(ROOT, "tests", "_build", "libclassic_client.so")


def build_client():
    if not os.path.exists(os.path.join(PKG, "libhavoc_classic.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(PKG, "csrc")])
    os.makedirs(os.path.dirname(CLIENT), exist_ok=True)
    src = os.path.join(ROOT, "oracle", "ref_shim.cpp")
    if not os.path.exists(CLIENT) or os.path.getmtime(CLIENT) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(PKG, "libhavoc_classic.so"))):
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-fPIC", "-shared", "-w", "-I" + os.path.join(ROOT, "include", "havoc"),
                               src, "-o", CLIENT, "-L" + PKG, "-lhavoc_classic", "-lhavoc_mi355x", "-Wl,-rpath," + PKG])
    return CLIENT


def _defined(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_reference_client_compiles_against_our_headers():
    assert os.path.exists(build_client())


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built")
def test_same_mangled_symbols_as_reference_library():
    build_client()
    ours = _defined(os.path.join(PKG, "libhavoc_classic.so"))
    ref = _defined(REF_SO)
    api = [s for s in ref if ("populate" in s.lower() or s in ("havoc_new_code", "havoc_delete_code", "havoc_get_ssd_linear",
                                                               "havoc_instruction_set_support", "havoc_main",
                                                               "havoc_print_instruction_set_support"))
           and "test" not in s.lower() and "populateAsm" not in s       # populateAsm: internal helper of pred_intra.cpp
           and not s.startswith("_ZN5havoc5intra8populateI")]           # havoc::intra::populate<bitDepth, Sample>: ditto (out of line at -O1)
    assert len(api) >= 30
    missing = sorted(s for s in api if s not in ours)
    assert not missing, missing


@pytest.mark.gpu
def test_classic_tables_match_golden_on_gpu():
    client = Reference(0, path=build_client())
    d, exp = load_golden()
    keys = ["u8.sad", "u16.sad", "u8.ssd", "u8.satd", "u16.satd", "u8.pred_uni", "u16.pred_uni", "u8.pred_bi", "u16.pred_bi",
            "subtract_bi", "u8.intra", "u16.intra", "u8.itx", "u16.itx", "fwd8", "fwd10", "quant", "qrec", "ssd_linear"]
    got = suite.run(suite.LoopImpl(client), d, keys=keys)
    assert len(got) >= 30
    for k in sorted(got):
        assert np.array_equal(got[k].astype(np.int64), exp[k].astype(np.int64)), k
