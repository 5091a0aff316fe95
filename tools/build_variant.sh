#!/bin/bash
# build a kernel-library variant into build_var/<name>.so with extra -D flags.   bash tools/build_variant.sh <name> [-DRR_X=1 ...]
name=$1; shift
mkdir -p build_var
cd brax-rodent-run_amd/csrc && hipcc --offload-arch=gfx950 -O2 -std=c++17 -shared -fPIC -fno-slp-vectorize -Xclang -target-feature -Xclang -load-store-opt "$@" -o ../../build_var/$name.so rr_api.hip 2>&1 | grep -v "not a recognized feature" | grep -v "^$" | head -20
