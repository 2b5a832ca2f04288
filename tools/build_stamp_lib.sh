#!/bin/bash
# Diagnostic library for tools/stamp_probe.py: the filter kernel compiled with -DRK_STAMP (cycle
# stamps around its phases), linked with the objects of the regular build.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/tools/_variants"
cd "$root/recommend_amd/csrc"
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off \
  -Wno-inline-asm -DRK_STAMP -c eval_filter.hip -o /tmp/ef_stamp.o
cd ../lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_variants/lib_stamp.so" \
  $(ls *.o | grep -v '^eval_filter.hip.o$') /tmp/ef_stamp.o
