#!/bin/bash
# Runs on the GPU box (via gpurun): bench.py plain, under rocprofv3 kernel-trace, and under two
# separate PMC passes; leaves everything under gpurun_out/<tag>/ for scripts/summarize_profile.py.
#   scripts/profile_bench.sh <tag> [bench args...]
set -uo pipefail
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py "$@" > "$out/bench.log" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
tail -1 "$out/bench.log"
# the kernel trace is taken of exactly the command that produced the bench line (the driver's: --gpus 1 --steps 20 --warmup 5)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 bench.py "$@" > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o f -- python3 bench.py --no-cpu-baseline --no-convergence --steps 2 "$@" > "$out/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$out/pmc_fetch.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o w -- python3 bench.py --no-cpu-baseline --no-convergence --steps 2 "$@" > "$out/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$out/pmc_write.log"; exit 1; }
find "$out" -name "*_kernel_stats.csv" -o -name "*_counter_collection.csv" | head
