#!/bin/bash
# GPU call U: workgroup shape of k_intra_satd35 for 16x16 / 32x32 partitions (HAVOC_INTRA35_VARIANT: 0 = 9 / 2 partitions x 256 threads,
# 1 = 4 / 1 x 128, 2 = 18 / 4 x 512): parity, then the isolated launch time.  (The switch was removed from kernels_fused.hip after this
# measurement -- no shape was faster; the commit before "experiments: k_intra_satd35 workgroup shapes measured" has it.)
tag=${1:-r02u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for v in 1 2 0; do
  HAVOC_INTRA35_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "golden or other_seed or reference_library" --timeout 500 -p no:cacheprovider > $O/${tag}_pytest_$v.log 2>&1
  echo "variant $v: $(tail -1 $O/${tag}_pytest_$v.log | cut -c1-200)"
  for bd in 8 10; do
    HAVOC_INTRA35_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --extra-4k 0 --bit-depth $bd --steps 60 --warmup 5 --kernel-reps 50 2> $O/${tag}_${bd}_$v.err | tail -1 > $O/${tag}_${bd}_$v.json
    python - <<PY
import json
r = json.load(open("$O/${tag}_${bd}_$v.json"))
k = r["whole_step"]["kernel_ms"]
print("variant $v bitDepth $bd:", r["value"], "fps; intra_satd35 ms", k.get("intra_satd35"), "checksum", r.get("checksum") or r["whole_step"].get("checksum"))
PY
  done
done
