#!/bin/bash
# Upper bound of XCD-resident coefficient / gradient-change rows for the binned pair (config 5): timing-only
# ablations of an experiments build -- the rows come from a 1 MB window (an L2 hit) instead of 12.8 / 16.8 MB tables
set -uo pipefail
out=$PWD/gpurun_out/r04_c5x
mkdir -p "$out"
EXTRA_FLAGS=-DSGDNET_EXPERIMENTS ./build.sh > "$out/build.log" 2>&1 || { tail -5 "$out/build.log"; exit 1; }
for ab in 0 32 64 96; do
  SGDNET_ABLATE=$ab timeout -k 10 400 python3 bench.py --workload C5s --steps 10 --warmup 2 --no-cpu-baseline --no-convergence > "$out/bench_$ab.json" 2> "$out/bench_$ab.err" || { tail -5 "$out/bench_$ab.err"; exit 1; }
  python3 - "$out/bench_$ab.json" $ab <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ablate", sys.argv[2], "ms/epoch", round(d["ms_per_step"], 3), "gather us", round(d["roofline"]["avg_launch_us"], 1), "sweep us", round(d["roofline"]["sweep_avg_launch_us"], 1))
PY
done
