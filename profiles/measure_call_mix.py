#!/usr/bin/env python3
"""The call mix of the REFERENCE'S OWN ENCODER by block size, measured on the CPU: oracle/_ref/turing_ref_classic (the reference encoder linked
against libhavoc_classic.so) over the stand-in device tests/mock_device.c, which tallies the jobs of every table call by entry point and block
size (HAVOC_MOCK_HISTOGRAM).  Prints / writes a JSON next to the mixes turingcodec_amd/workload.py ASSUMES for the synthetic picture (the PU-size
mix of the motion searches and the intra partition mix: SURVEY Appendix A.2 gives call counts, not sizes).  Test infrastructure: needs
/root/reference at build time (make -C oracle encoder) and no GPU.

    python profiles/measure_call_mix.py [case] [out.json]        case: a key of tests/encoder_tools.py CASES (default ra_medium_qp32)
"""
import collections
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import encoder_tools as et  # noqa: E402
import search_runner  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "ra_medium_qp32"
    out = sys.argv[2] if len(sys.argv) > 2 else None
    assert et.have_encoders(), "oracle/_ref/turing_ref_classic not built (make -C oracle encoder)"
    search_runner.build_mock()
    mock_dir = os.path.join(ROOT, "tests", "_build", "mock")
    with tempfile.TemporaryDirectory() as work:
        hist = os.path.join(work, "hist.json")
        et.encode(et.CLASSIC_EXE, case, work, env={"LD_LIBRARY_PATH": mock_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), "HAVOC_MOCK_HISTOGRAM": hist})
        raw = json.load(open(hist))
        # the searches themselves, counted by the reference encoder with trace points in its decision loops (oracle/trace_hooks.h; summary mode: counts only)
        searches = None
        trace_exe = os.path.join(et.REFDIR, "turing_ref_trace")
        if os.path.exists(trace_exe):
            summ = os.path.join(work, "summary.json")
            et.encode(trace_exe, case, work, env={"HAVOC_TRACE_SUMMARY": summ}, tag=".trace")
            searches = json.load(open(summ))
    by_fn = collections.defaultdict(dict)
    for k, v in raw.items():
        fn, size = k.split(" ")
        by_fn[fn][size] = v
    w, h, frames, seed, bd, opts = et.CASES[case]
    rep = {"case": case, "clip": f"{w}x{h}, {frames} frames, {bd}-bit, synthetic (seed {seed})", "options": opts, "calls_by_entry_point": {k: sum(v.values()) for k, v in by_fn.items()},
           "by_size": {k: dict(sorted(v.items(), key=lambda kv: -kv[1])) for k, v in by_fn.items()}}

    def share_by_area(d, key=lambda s: s):
        tot = sum(d.values())
        acc = collections.Counter()
        for size, n in d.items():
            acc[key(size)] += n
        return {k: round(v / tot, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}

    def cls(size):      # the classes the workload's PU mix is stated in
        a, b = (int(v) for v in size.split("x"))
        m = max(a, b)
        return f"max side {m}" + ("" if a == b else " (rectangular)")
    # every uni-directional search makes one single-position SAD at the zero vector and one or two at its predictors whatever its size: the
    # single-position SAD calls by size are the searched PUs' size mix
    if "sad" in by_fn:
        rep["searched_pu_size_mix_from_single_sad_calls"] = share_by_area(by_fn["sad"], cls)
    if "intra" in by_fn:      # 35 (or fewer) predictions per partition in the SATD stage + the refinements: shares by partition size
        rep["intra_prediction_calls_by_size"] = share_by_area(by_fn["intra"])
    if "transform" in by_fn:
        rep["forward_transform_calls_by_size"] = share_by_area(by_fn["transform"])
    if searches:
        k = {int(a): b for a, b in searches["records_by_kind"].items()}
        rep["searches"] = {"what": "counted inside the reference encoder's own loops (oracle/_ref/turing_ref_trace, summary mode)",
                           "searchMotionUni": k.get(1, 0), "searchMotionBi": k.get(9, 0), "havoc_sad calls": k.get(3, 0), "havoc_sad_multiref calls": k.get(4, 0),
                           "costDistortionMv calls (interpolate + SATD)": k.get(5, 0), "searchIntraPartition": k.get(12, 0),
                           "predictIntraLuma calls of the 35-mode stage": k.get(13, 0), "intra RD candidates (reconstructIntraLuma)": k.get(15, 0),
                           "uni_searches_by_size": searches["uni_searches_by_size"], "bi_searches_by_size": searches["bi_searches_by_size"],
                           "intra_partitions_by_log2_size": searches["intra_partitions_by_log2_size"]}
    from turingcodec_amd import workload
    tot = sum(n for _, _, n in workload.PU_MIX)
    rep["workload_assumes"] = {"pu_mix (w x h: share)": {f"{w}x{h}": round(n / tot, 4) for w, h, n in workload.PU_MIX},
                               "intra_mix (size: share)": {f"{1 << l}x{1 << l}": round(n / sum(m for _, m in workload.INTRA_MIX), 4) for l, n in workload.INTRA_MIX},
                               "tu_mix (size, dst: calls per 5-frame SURVEY A.2 sample)": {f"{1 << l}x{1 << l}{' DST' if t else ''}": n for l, t, n in workload.TU_MIX}}
    text = json.dumps(rep, indent=1)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
