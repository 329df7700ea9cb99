#!/bin/bash
# A/B on one box: the tree in ab_old/ (an earlier commit, built on the development box) against the current tree
set -uo pipefail
out=$PWD/gpurun_out/${1:-r04k}
mkdir -p "$out"
for round in 1 2; do
for which in old new; do
  if [ $which = old ]; then dir=ab_old; else dir=.; fi
  (cd $dir && timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > "$out/bench_${which}_$round.json" 2> "$out/bench_${which}_$round.err") || { tail -5 "$out/bench_${which}_$round.err"; exit 1; }
  python3 - "$out/bench_${which}_$round.json" $which <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "epochs/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "alone", round(d["roofline"].get("kernel_alone", {}).get("avg_launch_us", 0), 1))
PY
done
done
