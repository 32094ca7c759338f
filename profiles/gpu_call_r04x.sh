#!/bin/bash
# round 4, call x: the intra picture with running reconstruction against the reference's loop
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04x; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_intra_chain.py -x -q -m gpu -s 2>&1 | tail -25 > $O/pytest.log
