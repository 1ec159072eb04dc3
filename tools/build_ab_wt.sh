#!/bin/bash
# Build the WORKING TREE's engine with extra compiler flags into build_ab/lib_<name>.so (experiments behind -D switches):
#   tools/build_ab_wt.sh <name> [flags...]
set -e
name=$1; shift
mkdir -p build_ab
(cd rust_robotics_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm --offload-arch=gfx950 \
   -I../../include -shared "$@" -o ../../build_ab/lib_$name.so pf_engine.hip fs1_engine.hip selftest.hip -ldl -Wl,-rpath,/opt/rocm/lib)
echo "build_ab/lib_$name.so <- working tree $*"
