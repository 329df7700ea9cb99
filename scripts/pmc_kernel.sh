#!/bin/bash
# Hardware-counter passes over one bench.py workload, one rocprofv3 run per counter group
# (PMC only, no tracing):  scripts/pmc_kernel.sh <tag> "<bench args>" "<group 1>" "<group 2>" ...
set -uo pipefail
tag=$1; shift
bargs=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for group in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_$i" -o c -- python3 bench.py --no-cpu-baseline --no-convergence --steps 2 --warmup 1 $bargs > "$out/pmc_$i.log" 2>&1 || { echo "group $i failed"; tail -3 "$out/pmc_$i.log"; }
done
ls "$out"
