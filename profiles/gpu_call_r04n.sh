#!/bin/bash
# round 4, call n: merge candidates + chroma chain with the job tables made on the device
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_decisions.py -x -q -m gpu 2>&1 | tail -25
timeout 300 python bench.py --decisions 2 --decision-pictures 8 --res 1920x1080 2>&1 | tail -3
} > gpurun_out/r04n_call.log 2>&1
