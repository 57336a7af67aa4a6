#!/bin/bash
# Build a variant of the library for A/B timing: tools/ab_build.sh NAME [FILE=GITREV ...]
# copies plonk_amd/csrc to tools/ab_NAME (same depth, so relative includes hold), replaces
# each FILE by its content at GITREV and builds tools/bin/NAME.so with the library's own Makefile.
# AB_FLAGS in the environment adds compiler flags (e.g. -DPLK_NT_PASS_STORE=1).
# Time it with PLONK_HIP_LIB=tools/bin/NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=tools/ab_$name
rm -rf "$src"; mkdir -p "$src" tools/bin
cp plonk_amd/csrc/*.cuh plonk_amd/csrc/*.hip plonk_amd/csrc/*.cpp plonk_amd/csrc/*.h plonk_amd/csrc/Makefile "$src"/
for spec in "$@"; do
  f=${spec%%=*}; rev=${spec#*=}
  git show "$rev:plonk_amd/csrc/$f" > "$src/$f"
done
make -s -j3 -C "$src" OUT="../bin/$name.so" CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-function $AB_FLAGS"
echo "tools/bin/$name.so"
