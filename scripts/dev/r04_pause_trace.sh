#!/bin/bash
set -uo pipefail
out=$PWD/gpurun_out/${1:-r04g}
mkdir -p "$out"
export TMPDIR=/tmp
EXTRA_FLAGS=-DSGDNET_EXPERIMENTS ./build.sh > "$out/build.log" 2>&1 || { tail -5 "$out/build.log"; exit 1; }
for us in 100 400; do
export SGDNET_RNG_PAUSE_US=$us
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/trace$us" -o t -- python3 bench.py --no-cpu-baseline --no-convergence --steps 20 --warmup 5 > "$out/trace$us.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace$us.log"; exit 1; }
tail -1 "$out/trace$us.log" | cut -c1-200
done
