#!/bin/bash
# Build the engine of another git revision into build_ab/lib_<name>.so for tools/ab_bench.sh / RR_AMD_LIBRARY A/B runs:
#   tools/build_ab.sh <git-rev> <name>
set -e
rev=$1; name=$2
tmp=$(mktemp -d)
git archive "$rev" rust_robotics_amd/csrc include | tar -x -C "$tmp"
mkdir -p build_ab
(cd "$tmp/rust_robotics_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm --offload-arch=gfx950 \
   -I../../include -shared -o out.so pf_engine.hip fs1_engine.hip selftest.hip -ldl -Wl,-rpath,/opt/rocm/lib)
cp "$tmp/rust_robotics_amd/csrc/out.so" "build_ab/lib_$name.so"
rm -rf "$tmp"
echo "build_ab/lib_$name.so <- $rev"
