#!/usr/bin/env python3
"""Per-primitive table from a bench.py JSON line (the reference's CPU path next to the HIP kernels,
per call):  python profiles/per_primitive.py profiles/r01_bench.json > profiles/r01_per_primitive.md

CPU column: the reference's own havoc x86-JIT functions (oracle/_ref) on the same job tables, time per call on ONE core
(= per-frame time of the group x threads / calls).  GPU column: the batch kernel's isolated duration / calls."""
import json
import sys

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    from turingcodec_amd.workload import FrameWorkload
    w, h = (int(v) for v in r["config"]["workload"].split()[0].split("x"))
    bd = 8 if r["dtype"] == "u8" else 10
    wl = FrameWorkload(w, h, bd, 11)
    calls = {"sad4": len(wl.sad4), "sad": len(wl.sad), "pred_uni8": len(wl.uni8), "satd_inter": len(wl.satd_inter), "pred_uni4": len(wl.uni4),
             "pred_bi8": len(wl.bi8), "pred_bi4": len(wl.bi4), "subtract_bi": len(wl.subtract_bi),
             "subpel(interp+satd)": sum(len(j) for j in wl.subpel.values()), "intra": sum(len(j) for j in wl.intra.values()),
             "intra_satd35": 35 * sum(len(j) for j in wl.intra_search.values()),
             "tu_forward": sum(len(g["jobs"]) for g in wl.tu.values()), "tu_reconstruct": sum(len(g["jobs"]) for g in wl.tu.values())}
    cpu = r["cpu_baseline"]["ms_per_frame_by_group"]
    cores = r["cpu_baseline"]["cores"]
    gpu = dict(r["whole_step"]["kernel_ms"])
    gpu["subpel(interp+satd)"] = gpu.get("interp_planes", 0) + gpu.get("satd_planes", 0) + gpu.get("subpel_satd", 0)
    gpu["tu_reconstruct"] = gpu.get("tu_reconstruct", 0) + gpu.get("ssd", 0)
    print(f"Per-primitive cost, {r['config']['workload'].split(' random')[0]} (from {Path(sys.argv[1]).name})\n")
    print("| group (reference calls it stands for) | calls / frame | reference x86-JIT, ns per call on one core | MI355X batch kernel, ns per call | calls per second, 1 GPU vs 1 core |")
    print("|---|---|---|---|---|")
    note = {"subpel(interp+satd)": "sub-pel candidates: HavocPredUni + measureSatd", "tu_forward": "TU: residual + forward transform",
            "tu_reconstruct": "TU: de-quant + inverse transform + add + SSD", "intra_satd35": "intra mode: prediction + SATD"}
    for g in cpu:
        if g not in gpu or not calls.get(g):
            continue
        c_ns = cpu[g] * 1e6 * cores / calls[g]
        g_ns = gpu[g] * 1e6 / calls[g]
        print(f"| {g}{' (' + note[g] + ')' if g in note else ''} | {calls[g]} | {c_ns:.0f} | {g_ns:.2f} | {c_ns / g_ns:.0f}x |")
    print(f"\nWhole frame: reference {1e3 / r['cpu_baseline']['value']:.1f} ms on {cores} threads, MI355X {r['ms_per_step']:.3f} ms.")


if __name__ == "__main__":
    main()
