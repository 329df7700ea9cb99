#!/bin/bash
# C5 epoch: plain stores (product) against write-through stores in the sweep (swt) and in sweep + gather (gwt)
set -uo pipefail
out=$PWD/gpurun_out/r04_c5wt
mkdir -p "$out"
for v in plain swt gwt plain; do
  if [ $v == plain ]; then unset SGDNET_LIB_PATH; else export SGDNET_LIB_PATH=$PWD/build/variants/libsgdnet_hip_$v.so; fi
  timeout -k 10 600 python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-convergence > "$out/bench_$v.json" 2> "$out/bench_$v.err" || { echo "$v failed"; tail -3 "$out/bench_$v.err"; exit 1; }
  python3 - "$out/bench_$v.json" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "C5", round(d["value"], 2), "epochs/s", round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "us; sweep", round(d["roofline"].get("sweep_avg_launch_us"), 1))
PY
done
