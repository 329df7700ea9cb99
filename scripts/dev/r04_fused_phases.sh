#!/bin/bash
# fused epoch kernel: bench line, kernel trace (side-stream kernels beside it), in-kernel phase times
set -uo pipefail
tag=${1:-r04b}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_fused.json" 2> "$out/bench_fused.err" || { tail -5 "$out/bench_fused.err"; exit 1; }
tail -c 1500 "$out/bench_fused.json"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 bench.py --no-cpu-baseline --no-convergence --steps 20 --warmup 5 > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
EXTRA_FLAGS=-DSGDNET_PHASE_TIMING ./build.sh > "$out/build_phase.log" 2>&1 || { tail -5 "$out/build_phase.log"; exit 1; }
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > "$out/bench_phase.json" 2> "$out/bench_phase.err" || { tail -5 "$out/bench_phase.err"; exit 1; }
grep "phase" "$out/bench_phase.err"
