#!/bin/bash
# C5 epoch once more (after the binned gather lost its in-loop scratch reloads): kernel trace + bench line
set -uo pipefail
out=$PWD/gpurun_out/r04_c5b
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5" -o t -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > "$out/trace_c5.log" 2>&1 || { echo "trace c5 failed"; tail -3 "$out/trace_c5.log"; exit 1; }
tail -1 "$out/trace_c5.log" > "$out/bench_c5.json"
python3 - "$out/bench_c5.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C5", round(d["value"], 2), "epochs/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "us; sweep", d["roofline"].get("sweep_avg_launch_us"), "frac", round(d["roofline"]["frac"], 4))
PY
f=$(find "$out/trace_c5" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" "$out/c5_kernel_stats.csv"; head -6 "$f" | cut -c1-170; }
