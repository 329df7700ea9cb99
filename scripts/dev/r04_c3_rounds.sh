#!/bin/bash
# C3, in-kernel phase times of the fused epoch kernel with 1, 2 and 4 rounds per epoch (window 125000 / 62500 / 31250):
# which part of a phase's time is per round and which is per launch?
set -uo pipefail
out=$PWD/gpurun_out/r04_c3_rounds
mkdir -p "$out"
EXTRA_FLAGS=-DSGDNET_PHASE_TIMING ./build.sh > "$out/build_phase.log" 2>&1 || { tail -5 "$out/build_phase.log"; exit 1; }
for b in 125000 62500 31250; do
  timeout -k 10 300 python3 bench.py --workload C3 --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > "$out/bench_$b.json" 2> "$out/bench_$b.err" || { tail -5 "$out/bench_$b.err"; exit 1; }
  echo "== window $b"; grep "phase" "$out/bench_$b.err" | head -10
done
