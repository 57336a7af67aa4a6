#!/bin/bash
# Build a variant of the library for A/B timing: tools/ab_build.sh NAME [FILE=GITREV ...]
# copies plonk_amd/csrc to tools/ab_NAME (same depth, so relative includes hold), replaces
# each FILE by its content at GITREV and writes tools/bin/NAME.so.
# AB_FLAGS in the environment adds compiler flags (e.g. -DPLK_BIG_SH_MEM=1).
# Time it with PLONK_HIP_LIB=tools/bin/NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=tools/ab_$name
rm -rf "$src"; mkdir -p "$src"
cp plonk_amd/csrc/*.cuh plonk_amd/csrc/capi.hip plonk_amd/csrc/node.hip "$src"/
for spec in "$@"; do
  f=${spec%%=*}; rev=${spec#*=}
  git show "$rev:plonk_amd/csrc/$f" > "$src/$f"
done
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $AB_FLAGS -shared -o "tools/bin/$name.so" "$src/capi.hip" "$src/node.hip"
echo "tools/bin/$name.so"
