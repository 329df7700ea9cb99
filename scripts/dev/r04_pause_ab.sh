#!/bin/bash
# A/B of the pause in front of the generators (experiments build: SGDNET_RNG_PAUSE_US)
set -uo pipefail
out=$PWD/gpurun_out/${1:-r04f}
mkdir -p "$out"
EXTRA_FLAGS=-DSGDNET_EXPERIMENTS ./build.sh > "$out/build.log" 2>&1 || { tail -5 "$out/build.log"; exit 1; }
for us in 0 10 40 100; do
  SGDNET_RNG_PAUSE_US=$us timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-convergence > "$out/bench_$us.json" 2> "$out/bench_$us.err" || { tail -5 "$out/bench_$us.err"; exit 1; }
  python3 - "$out/bench_$us.json" $us <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pause", sys.argv[2], "epochs/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "kernel us", round(d["roofline"]["avg_launch_us"], 1))
PY
done
