#!/bin/bash
# the timing variant of libhavoc_mi355x.so: kernels_rdoq.hip with -DHAVOC_RDOQ_TIMING, every other object as built by csrc/Makefile
set -e
R=$(cd $(dirname $0)/../.. && pwd); cd $R/turingcodec_amd/csrc
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DHAVOC_RDOQ_TIMING -c kernels_rdoq.hip -o /tmp/kernels_rdoq_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/micro/libhavoc_mi355x_timing.so $(ls *.o | grep -v kernels_rdoq.o) /tmp/kernels_rdoq_timing.o
