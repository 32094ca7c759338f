#!/bin/bash
# surfaces of range 72 / 96 against the SAD definition, then call G
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surface" 2>&1 | tail -2
bash profiles/r03/gpu_call_g.sh
