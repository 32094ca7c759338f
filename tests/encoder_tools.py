"""TEST INFRASTRUCTURE: running the reference's own encoder (oracle/_ref/turing_ref_*, built by `make -C oracle encoder` from the
sources under /root/reference where they lie + oracle/ref_encoder_main.cpp) on seeded synthetic clips.

  turing_ref_havoc    the reference encoder over the reference's havoc library (x86 JIT tables with --asm 1, plain C with --asm 0)
  turing_ref_classic  the SAME encoder objects over turingcodec_amd/libhavoc_classic.so, i.e. every table call of the encode
                      goes to the MI355X library (or, in the CPU suite, to the stand-in device tests/mock_device.c)

The clips are the SURVEY 8(d) generator (turingcodec_amd.workload.synth_frames); tests/golden/encoder_streams.json holds the MD5 of
each clip and of the stream the reference encoder writes for it (generated here by tests/golden/make_encoder_golden.py).
"""
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HAVOC_EXE = os.path.join(REFDIR, "turing_ref_havoc")
CLASSIC_EXE = os.path.join(REFDIR, "turing_ref_classic")
HOOKED_EXE = os.path.join(REFDIR, "turing_ref_hooked")      # turing_ref_classic + the two havoc_classic_register_picture calls (oracle/register_hooks.h)
GOLDEN = os.path.join(ROOT, "tests", "golden", "encoder_streams.json")

# name: (width, height, frames, clip seed, bit depth, encoder options).  --no-sao everywhere: with SAO the reference is not
# deterministic.  Geometry with partial CTUs on both edges.
CASES = {
    # BASELINE configs[1]-like: random access, speed=medium (RDOQ, RQT, AMP, bi-prediction)
    "ra_medium_qp32": (416, 240, 9, 7, 8, ["--qp", "32", "--speed", "medium"]),
    "ra_medium_qp22": (416, 240, 5, 7, 8, ["--qp", "22", "--speed", "medium"]),
    "ra_fast_qp32": (416, 240, 9, 7, 8, ["--qp", "32", "--speed", "fast"]),
    "ra_slow_qp27": (256, 144, 3, 9, 8, ["--qp", "27", "--speed", "slow"]),
    # configs[0]: all-intra speed=fast (havoc_quantize in the chain, no RDOQ)
    "ai_fast_qp32": (416, 240, 3, 7, 8, ["--qp", "32", "--speed", "fast", "--max-gop-n", "1", "--max-gop-m", "1"]),
    # configs[3]-like: Main10, 16-bit sample tables
    "ra_medium_10bit_qp27": (416, 240, 5, 17, 10, ["--qp", "27", "--speed", "medium", "--bit-depth", "10"]),
    "ra_medium_internal10": (256, 144, 5, 9, 8, ["--qp", "32", "--speed", "medium", "--internal-bit-depth", "10"]),
    # short ones for the GPU box (every table call is a launch there: ~1 M calls per case)
    "gpu_ra_medium_qp32": (416, 240, 3, 7, 8, ["--qp", "32", "--speed", "medium"]),
    "gpu_ra_medium_qp22": (256, 144, 3, 9, 8, ["--qp", "22", "--speed", "medium"]),
    "gpu_ai_fast_qp32": (256, 144, 1, 9, 8, ["--qp", "32", "--speed", "fast", "--max-gop-n", "1", "--max-gop-m", "1"]),
    "gpu_ra_medium_10bit": (256, 144, 3, 21, 10, ["--qp", "27", "--speed", "medium", "--bit-depth", "10"]),
    # not a stream test: the bench's picture size and clip generator, for profiles/measure_call_mix.py (the reference's call mix by block size)
    "mix_1080p_qp32": (1920, 1080, 9, 11, 8, ["--qp", "32", "--speed", "medium"]),
    "mix_4k_qp32": (3840, 2160, 9, 11, 8, ["--qp", "32", "--speed", "medium"]),
}


def have_encoders():
    return os.path.exists(HAVOC_EXE) and os.path.exists(CLASSIC_EXE)


def md5(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return hashlib.md5(path_or_bytes).hexdigest()
    with open(path_or_bytes, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def write_clip(path, width, height, frames, seed, bit_depth):
    from turingcodec_amd import workload
    with open(path, "wb") as f:
        for planes in workload.synth_frames(width, height, frames, seed, bit_depth=bit_depth):
            for p in planes:
                f.write(np.ascontiguousarray(p).tobytes())
    return path


def encode(exe, case, workdir, extra=(), env=None, timeout=900, tag=""):
    """runs one encode, returns (stream bytes, stderr text)"""
    w, h, n, seed, bd, opts = CASES[case]
    clip = os.path.join(workdir, f"{case}.yuv")
    if not os.path.exists(clip):
        write_clip(clip, w, h, n, seed, bd)
    out = os.path.join(workdir, f"{case}{tag}.{os.path.basename(exe)}.bit")
    cmd = [exe, "--input-res", f"{w}x{h}", "--frames", str(n), "--frame-rate", "24", "--verbosity", "0", "--no-sao"] + list(opts) + list(extra) + ["-o", out, clip]
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, f"{' '.join(cmd)} failed ({r.returncode}): {r.stderr[-2000:]}"
    with open(out, "rb") as f:
        return f.read(), r.stderr


def golden():
    with open(GOLDEN) as f:
        return json.load(f)
