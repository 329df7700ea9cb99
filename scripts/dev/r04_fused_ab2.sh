#!/bin/bash
set -uo pipefail
tag=${1:-r04j}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
for f in 1 2 3 1 2 3; do
timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-convergence --fused-epoch $f > "$out/bench_$f.json" 2> "$out/bench_$f.err" || { tail -5 "$out/bench_$f.err"; exit 1; }
python3 - "$out/bench_$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fused_epoch", sys.argv[2], "epochs/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "alone", round(d["roofline"].get("kernel_alone", {}).get("avg_launch_us", 0), 1))
PY
done
