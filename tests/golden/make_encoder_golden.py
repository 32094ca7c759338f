"""Generates tests/golden/encoder_streams.json: for every case of tests/encoder_tools.py the MD5 of the synthetic clip and of the
stream the REFERENCE encoder (oracle/_ref/turing_ref_havoc = /root/reference's own sources, `make -C oracle encoder`) writes for it, with
its x86 JIT tables (--asm 1) -- after checking that the plain-C tables (--asm 0) and one worker thread give the same stream
(the reference's own signature test asserts exactly that, turing/signature.cpp:231-233).  Data only: hashes and sizes."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import encoder_tools as et  # noqa: E402

out = {}
with tempfile.TemporaryDirectory() as d:
    for case in et.CASES:
        jit, _ = et.encode(et.HAVOC_EXE, case, d, ["--asm", "1"])
        c, _ = et.encode(et.HAVOC_EXE, case, d, ["--asm", "0"], tag=".c")
        one, _ = et.encode(et.HAVOC_EXE, case, d, ["--asm", "1", "--threads", "1"], tag=".t1")
        assert jit == c == one, case
        out[case] = {"clip_md5": et.md5(os.path.join(d, case + ".yuv")), "stream_md5": et.md5(jit), "stream_bytes": len(jit)}
        print(case, out[case])
with open(et.GOLDEN, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
