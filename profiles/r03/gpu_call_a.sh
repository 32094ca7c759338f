#!/bin/bash
# round 3, GPU call A: the reference's own encoder over the MI355X tables (stream identical to the CPU reference encoder), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
nproc > $O/nproc.txt
( time timeout 1500 python -m pytest tests/test_reference_encoder.py -m gpu -x -q -s ) > $O/encoder_gpu.log 2>&1
tail -15 $O/encoder_gpu.log
( time timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_reference_encoder.py ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
