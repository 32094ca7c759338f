"""Generates tests/golden/sao_golden.npz: outputs of the REFERENCE's SAO statistics templates (turing/EncSao.h) and filters
(turing/sao.cpp), compiled into oracle/_ref, on the seeded cases of tests/sao_tools.py.  python tests/golden/make_sao_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflibs       # noqa: E402
import sao_tools     # noqa: E402

N = 60


def main():
    ref = reflibs.Reference()
    out = {}
    for seed in range(N):
        c = sao_tools.make_case(seed)
        out[f"stats{seed}"] = ref.sao_stats(c["src"], c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], c["bd"])
        for kind, offs in ((1, c["band"]), (2, c["edge"][:5])):
            d = np.zeros_like(c["rec"])
            ref.sao_filter(d, c["origin"], c["stride"], c["rec"], c["origin"], c["stride"], c["w"], c["h"], kind, c["eo_class"], offs, c["bd"])
            out[f"filter{kind}_{seed}"] = d
    np.savez_compressed(os.path.join(HERE, "sao_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
