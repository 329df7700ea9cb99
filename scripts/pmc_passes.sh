#!/bin/bash
# Hardware-counter passes over bench.py (C4, N=1) for the dominant kernel; one rocprofv3 run per
# counter group (PMC only, no tracing).  Output: gpurun_out/<tag>/pmc_<i>/ CSVs.
set -uo pipefail
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for group in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_$i" -o c -- python3 bench.py --no-cpu-baseline --no-convergence --steps 2 > "$out/pmc_$i.log" 2>&1 || { echo "group $i failed"; tail -3 "$out/pmc_$i.log"; }
done
ls "$out"
