"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol the header
declares, and refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "havoc_mi355x.h")


@pytest.fixture(scope="module")
def lib():
    import turingcodec_amd
    if not os.path.exists(turingcodec_amd.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "turingcodec_amd", "csrc")])
    return C.CDLL(turingcodec_amd.LIB_PATH)


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(havoc_mi355x_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/havoc_mi355x.h but not exported"


def test_binding_covers_header():
    import turingcodec_amd
    assert set(declared_symbols()) == set(turingcodec_amd.exported_symbols())


def test_job_struct_sizes():
    """the sizes the kernels index jobs with (static_asserts in csrc/api.hip mirror this)"""
    text = open(HEADER).read()
    sizes = dict(re.findall(r"\}\s*(havoc_mi355x_[a-z0-9_]+_job);\s*/\*\s*(\d+) bytes", text))
    assert sizes == {"havoc_mi355x_pair_job": "16", "havoc_mi355x_sad4_job": "32", "havoc_mi355x_surface_job": "32", "havoc_mi355x_satd_multi_job": "80", "havoc_mi355x_pred_uni_job": "32",
                     "havoc_mi355x_pred_bi_job": "48", "havoc_mi355x_subtract_bi_job": "32",
                     "havoc_mi355x_intra_job": "32", "havoc_mi355x_tu_job": "16", "havoc_mi355x_quant_job": "32",
                     "havoc_mi355x_intra_search_job": "32", "havoc_mi355x_tu_fused_job": "16", "havoc_mi355x_rdoq_job": "48", "havoc_mi355x_sao_stats_job": "16", "havoc_mi355x_sao_chroma_job": "32", "havoc_mi355x_sao_job": "96"}


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib.havoc_mi355x_last_error.restype = C.c_char_p
    h = C.c_void_p()
    rc = lib.havoc_mi355x_create(C.byref(h), 0, None)
    assert rc != 0 and not h.value
    assert b"no CPU path" in lib.havoc_mi355x_last_error() or b"HIP" in lib.havoc_mi355x_last_error()
    import turingcodec_amd
    with pytest.raises(turingcodec_amd.HavocError):
        turingcodec_amd.Havoc(0)


def test_product_does_not_touch_oracle():
    """the product sources never include, link or import anything under oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "turingcodec_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                s = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle|reflibs|liboracle|libhavoc_ref", s):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/havoc_mi355x.h compiles as C99 with -Wall -Wextra -pedantic -Werror (no C++, no HIP, no torch types in
    the boundary), and a C client that names every declared function links against libhavoc_mi355x.so"""
    import turingcodec_amd
    names = declared_symbols()
    src = tmp_path / "client.c"
    src.write_text('#include "havoc_mi355x.h"\n#include <stddef.h>\n'
                   "typedef void (*fn)(void);\n"
                   "size_t sizes[] = {sizeof(havoc_mi355x_pair_job), sizeof(havoc_mi355x_sad4_job), sizeof(havoc_mi355x_tu_fused_job)};\n"
                   "fn table[] = {" + ", ".join(f"(fn){n}" for n in names) + "};\n"
                   "int main(void) { return sizeof table / sizeof table[0] == " + str(len(names)) + " ? 0 : 1; }\n")
    exe = tmp_path / "client"
    libdir = os.path.dirname(turingcodec_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lhavoc_mi355x", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    assert subprocess.call([str(exe)]) == 0
