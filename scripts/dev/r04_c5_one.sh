#!/bin/bash
# one C5 bench line (3 epochs), SGDNET_TRACE for the range count
set -uo pipefail
out=$PWD/gpurun_out/r04_c5one
mkdir -p "$out"
SGDNET_TRACE=1 timeout -k 10 600 python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > "$out/bench.json" 2> "$out/bench.err" || { echo "failed"; tail -3 "$out/bench.err"; exit 1; }
grep "feature ranges" "$out/bench.err" | head -2
python3 - "$out/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C5", round(d["value"], 2), "epochs/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "us; sweep", d["roofline"].get("sweep_avg_launch_us"), "frac", round(d["roofline"]["frac"], 4))
PY
