#!/bin/bash
# Builds libsgdnet_hip variants that differ in the flags of ONE source file, side by side under
# build/variants/ (development aid for A/B runs on one GPU box via SGDNET_LIB_PATH).
#   scripts/dev/build_variants.sh <file.hip> "name1:-DFLAG=.." "name2:-D.."
set -euo pipefail
cd "$(dirname "$0")/../.."
src=$1; shift
./build.sh > /dev/null
mkdir -p build/variants
FLAGS=$(cat build/.flags)
pids=()
for v in "$@"; do
  name=${v%%:*}; fl=${v#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $fl -x hip -c sgdnet_amd/csrc/$src -o build/variants/$name.o &&
    objs=""; for f in saga_exact saga_batched r_rng_device setup_device score solver driver r_rng mt_jump; do
      if [ "$f.hip" == "$src" ] || [ "$f.cpp" == "$src" ]; then objs="$objs build/variants/$name.o"; else objs="$objs build/$f.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libsgdnet_hip_$name.so $objs ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la build/variants/*.so
