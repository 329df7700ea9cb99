#!/bin/bash
# Builds sgdnet_amd/lib/libsgdnet_hip.so (gfx950 only) and the CPU oracle.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC=sgdnet_amd/csrc
OUT=sgdnet_amd/lib
mkdir -p "$OUT" build
# -ffp-contract=off: the exact-order kernels follow the reference's arithmetic
# order without fused multiply-adds.
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -I$SRC -Wall -Wno-unused-function ${EXTRA_FLAGS:-}"
# objects are rebuilt when the flags change (e.g. EXTRA_FLAGS=-DSGDNET_PHASE_TIMING experiments)
if [ ! -f build/.flags ] || [ "$(cat build/.flags)" != "$FLAGS" ]; then rm -f build/*.o; echo "$FLAGS" > build/.flags; fi
pids=()
for f in saga_exact.hip saga_batched.hip r_rng_device.hip setup_device.hip score.hip solver.cpp driver.cpp r_rng.cpp mt_jump.cpp; do
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ] || [ "$SRC/common.hpp" -nt "$o" ] || [ "$SRC/device_math.hpp" -nt "$o" ] || [ "$SRC/setup_device.hpp" -nt "$o" ] || [ include/sgdnet_hip.h -nt "$o" ] || [ include/sgdnet_detmath.h -nt "$o" ]; then
    $HIPCC $FLAGS -x hip -c "$SRC/$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsgdnet_hip.so" build/saga_exact.o build/saga_batched.o build/r_rng_device.o build/setup_device.o build/score.o build/solver.o build/driver.o build/r_rng.o build/mt_jump.o
make -s -C oracle liboracle.so liboracle_det.so
# the .Call shim (shim/sgdnet_shim.c) compiled as it will be inside the R package, against the
# mock of the R C API under tests/rmock (no R in this image): test infrastructure
gcc -O2 -std=gnu11 -Wall -Wextra -Wno-cast-function-type -fPIC -shared -Itests/rmock/include -Iinclude -o tests/rmock/libsgdnet_shim_mock.so \
    shim/sgdnet_shim.c tests/rmock/rmock.c -L"$OUT" -lsgdnet_hip -Wl,-rpath,'$ORIGIN/../../sgdnet_amd/lib' 
# ... and once more with the sanitizers on (CPU test of the shim's error paths: tests/test_shim_sanitizers.py)
gcc -O1 -g -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -Wall -Wextra -Wno-cast-function-type \
    -fPIC -shared -Itests/rmock/include -Iinclude -o tests/rmock/libsgdnet_shim_mock_asan.so \
    shim/sgdnet_shim.c tests/rmock/rmock.c -L"$OUT" -lsgdnet_hip -Wl,-rpath,'$ORIGIN/../../sgdnet_amd/lib'
echo "built $OUT/libsgdnet_hip.so"
