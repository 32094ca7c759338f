"""turingcodec_amd: MI355X (gfx950) implementation of the `havoc` primitive layer of the Turing HEVC encoder.

The product is the C-ABI shared library ``libhavoc_mi355x.so`` (sources: turingcodec_amd/csrc, header:
include/havoc_mi355x.h); this package is the thin host binding used by tests and bench.py.
"""
from .havoc import Havoc, HavocError, LIB_PATH, exported_symbols  # noqa: F401
