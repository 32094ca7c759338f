#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04z; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -X faulthandler bench.py --decisions 3 --gpus 1 --exchange --res 416x240 --pictures 17 > $O/w1.out 2> $O/w1.err; echo "rc=$?" >> $O/w1.err
tail -c 3000 $O/w1.err; tail -c 600 $O/w1.out
