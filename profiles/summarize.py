#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>/**/*.csv) into the small summaries committed under profiles/.

    python profiles/summarize.py r01 gpurun_out/prof_r01 gpurun_out/pmc_fetch_r01 gpurun_out/pmc_write_r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, havoc kernels only) and
profiles/<tag>_hbm_traffic.csv (per kernel: launches, mean FETCH_SIZE / WRITE_SIZE per launch in bytes with the
gfx950 correction of guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per 128-B request -> x2).
"""
import glob
import os
import sys

import pandas as pd


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return f[0] if f else None


def short(name):
    name = name.replace("void ", "").replace("havoc_gpu::", "")
    return name.split("(")[0]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    st = pd.read_csv(find(stats_dir, "kernel_stats.csv"))
    st = st[st["Name"].str.contains("havoc_gpu")].copy()
    st["Name"] = st["Name"].map(short)
    st.to_csv(os.path.join(here, f"{tag}_kernel_stats.csv"), index=False)
    print(st.to_string(index=False))
    rows = {}
    for d, col, mult in ((sys.argv[3] if len(sys.argv) > 3 else None, "FETCH_SIZE", 2.0),
                         (sys.argv[4] if len(sys.argv) > 4 else None, "WRITE_SIZE", 1.0)):
        if not d:
            continue
        c = pd.read_csv(find(d, "counter_collection.csv"))
        c = c[c["Kernel_Name"].str.contains("havoc_gpu") & (c["Counter_Name"] == col)].copy()
        c["Kernel"] = c["Kernel_Name"].map(short)
        g = c.groupby("Kernel")["Counter_Value"]
        for k, v in g.mean().items():
            rows.setdefault(k, {})[col + "_bytes_per_launch"] = v * 1024.0 * mult   # counters are in KiB
            rows[k]["launches_" + col] = int(g.count()[k])
    if rows:
        t = pd.DataFrame.from_dict(rows, orient="index").sort_index()
        t.index.name = "Kernel"
        t.to_csv(os.path.join(here, f"{tag}_hbm_traffic.csv"))
        print(t.to_string())


if __name__ == "__main__":
    main()
