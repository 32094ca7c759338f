#!/bin/bash
# GPU call G: RDOQ in the timed chain -- bench with and without it, isolated kernel durations, SQ counters of k_rdoq.
tag=${1:-r02g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="python $R/bench.py --extra-4k 0"
timeout 400 $B --rdoq 1 2> $O/${tag}_rdoq1.err | tail -1 > $O/${tag}_rdoq1.json
timeout 300 $B --rdoq 0 --no-cpu-baseline 2> $O/${tag}_rdoq0.err | tail -1 > $O/${tag}_rdoq0.json
python - <<PY
import json
for t in ("rdoq1", "rdoq0"):
    try:
        r = json.load(open("$O/${tag}_%s.json" % t))
        print(t, "fps", r["value"], "ms/step", r["ms_per_step"], "dominant", r["roofline"]["kernel"], r["roofline"]["frac"])
        print("  kernel_ms", r["whole_step"]["kernel_ms"])
        if "cpu_baseline" in r and r["cpu_baseline"]:
            c = r["cpu_baseline"]
            print("  cpu", c["value"], c.get("ms_per_frame_by_group"), (c.get("parity_vs_reference") or {}).get("compared"), (c.get("parity_vs_reference") or {}).get("mismatches"), (c.get("parity_vs_reference") or {}).get("mismatches_by_group"))
    except Exception as e:
        print(t, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_serial -- python $R/bench.py --no-cpu-baseline --extra-4k 0 --min-seconds 0 --steps 10 --warmup 2 --lanes 1 --no-graph > $O/${tag}_serial_bench.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${tag}_sq1 -- python $R/bench.py --no-cpu-baseline --extra-4k 0 --min-seconds 0 --steps 2 --warmup 1 --kernel-reps 1 --lanes 1 --no-graph > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/${tag}_sq2 -- python $R/bench.py --no-cpu-baseline --extra-4k 0 --min-seconds 0 --steps 2 --warmup 1 --kernel-reps 1 --lanes 1 --no-graph > /dev/null 2>&1
cd $R
f=$(find $O/${tag}_serial -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-200
