#!/bin/bash
set -uo pipefail
out=$PWD/gpurun_out/${1:-r04p}
mkdir -p "$out"
EXTRA_FLAGS=-DSGDNET_PHASE_TIMING ./build.sh > "$out/build_phase.log" 2>&1 || { tail -5 "$out/build_phase.log"; exit 1; }
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-convergence ${2:-} > "$out/bench_phase.json" 2> "$out/bench_phase.err" || { tail -5 "$out/bench_phase.err"; exit 1; }
grep "phase" "$out/bench_phase.err" | head -24
