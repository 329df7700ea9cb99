#!/bin/bash
# C5 epoch: binned gather with 8 (product) against 16 coefficient rows requested together (build/variants/libsgdnet_hip_w16.so)
set -uo pipefail
out=$PWD/gpurun_out/r04_c5w
mkdir -p "$out"
for v in w8 w16; do
  if [ $v == w16 ]; then export SGDNET_LIB_PATH=$PWD/build/variants/libsgdnet_hip_w16.so; else unset SGDNET_LIB_PATH; fi
  timeout -k 10 600 python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > "$out/bench_$v.json" 2> "$out/bench_$v.err" || { echo "$v failed"; tail -3 "$out/bench_$v.err"; exit 1; }
  python3 - "$out/bench_$v.json" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "C5", round(d["value"], 2), "epochs/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "us; sweep", d["roofline"].get("sweep_avg_launch_us"), "frac", round(d["roofline"]["frac"], 4))
PY
done
