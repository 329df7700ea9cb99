#!/bin/bash
set -uo pipefail
out=$PWD/gpurun_out/r04_c3c5
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --workload C3 --steps 20 --warmup 5 > "$out/bench_c3.json" 2> "$out/bench_c3.err" || { tail -5 "$out/bench_c3.err"; exit 1; }
python3 - "$out/bench_c3.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C3", round(d["value"], 1), "epochs/s", round(d["ms_per_step"], 4), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "alone", round(d["roofline"].get("kernel_alone", {}).get("avg_launch_us", 0), 1), "conv", d.get("convergence", {}).get("epochs"))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c3" -o t -- python3 bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline --no-convergence > "$out/trace_c3.log" 2>&1 || { echo "trace c3 failed"; tail -3 "$out/trace_c3.log"; }
f=$(find "$out/trace_c3" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-170
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5" -o t -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > "$out/trace_c5.log" 2>&1 || { echo "trace c5 failed"; tail -3 "$out/trace_c5.log"; exit 1; }
tail -1 "$out/trace_c5.log" | cut -c1-600
f=$(find "$out/trace_c5" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-170
