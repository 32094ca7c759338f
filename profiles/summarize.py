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
    name = name.replace("void ", "").replace("havoc_gpu::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def sq_summary(tag, dirs):
    """per kernel: mean SQ counters per launch and the derived occupancy figures the DESIGN.md bound discussion quotes
    (the SQ counters are summed over all SIMDs/CUs of the device)"""
    here = os.path.dirname(os.path.abspath(__file__))
    frames = []
    for d in dirs:
        c = pd.read_csv(find(d, "counter_collection.csv"))
        c = c[c["Kernel_Name"].str.contains("havoc_gpu")].copy()
        c["Kernel"] = c["Kernel_Name"].map(short)
        frames.append(c.pivot_table(index="Kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean"))
    t = pd.concat(frames, axis=1)
    # derived figures with rocprof's own formulas (VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE,
    # LDSBankConflict = 100 * SQ_LDS_BANK_CONFLICT / GRBM_GUI_ACTIVE / CU_NUM); the reported counter values are sums
    # over the 8 XCDs, so GRBM_GUI_ACTIVE is divided by 8 to get the kernel's busy cycles (checks against the
    # kernel-trace duration x 2.4 GHz)
    if {"SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"} <= set(t.columns):
        cyc = t["GRBM_GUI_ACTIVE"] / 8.0
        t["busy_us_at_2.4GHz"] = cyc / 2400.0
        t["VALUBusy_pct"] = 100.0 * 4 * t["SQ_ACTIVE_INST_VALU"] / 1024.0 / cyc
        t["LDS_issue_pct"] = 100.0 * 4 * t["SQ_ACTIVE_INST_LDS"] / 1024.0 / cyc
        t["LDSBankConflict_pct"] = 100.0 * t["SQ_LDS_BANK_CONFLICT"] / cyc / 256.0
    if {"SQ_INSTS_VALU", "SQ_WAVES"} <= set(t.columns):
        t["valu_insts_per_wave"] = t["SQ_INSTS_VALU"] / t["SQ_WAVES"].clip(lower=1)
        t["lds_insts_per_wave"] = t["SQ_INSTS_LDS"] / t["SQ_WAVES"].clip(lower=1)
    t.round(2).to_csv(os.path.join(here, f"{tag}_sq_counters.csv"))
    print(t.round(2).to_string())


def main():
    if sys.argv[1] == "--sq":
        return sq_summary(sys.argv[2], sys.argv[3:])
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
