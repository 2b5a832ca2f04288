#!/bin/bash
# tools/exp_build.sh NAME -DFLAG...: build a variant library with eval_filter.hip compiled under the flags
set -e
name=$1; shift
cd /root/repo/recommend_amd/csrc
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wno-inline-asm "$@" -c eval_filter.hip -o /tmp/ef_$name.o
cd ../lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_variants/lib_$name.so host.cpp.o graph.hip.o spmm.hip.o dense.hip.o glue.hip.o bpr.hip.o eval.hip.o proj.hip.o /tmp/ef_$name.o
