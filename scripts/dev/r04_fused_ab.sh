#!/bin/bash
# fused epoch kernel: parity tests, then A/B of the hand-off flavours (option fused_epoch 1 / 2) and the separate launches (0)
set -uo pipefail
tag=${1:-r04i}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_frontend.py -x -q -m gpu -k "virtual_shards or compact_records or handles_repeats or full_size or kkt or generators or stream or fit_" > "$out/pytest.log" 2>&1
rc=$?; tail -4 "$out/pytest.log"; [ $rc -ne 0 ] && exit 1
for f in 1 2 0 1; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-epoch $f > "$out/bench_$f.json" 2> "$out/bench_$f.err" || { tail -5 "$out/bench_$f.err"; exit 1; }
python3 - "$out/bench_$f.json" $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fused_epoch", sys.argv[2], "epochs/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "alone", round(d["roofline"].get("kernel_alone", {}).get("avg_launch_us", 0), 1), "conv", d.get("convergence", {}).get("epochs"), d.get("convergence", {}).get("deviance"))
PY
done
EXTRA_FLAGS=-DSGDNET_PHASE_TIMING ./build.sh > "$out/build_phase.log" 2>&1 || { tail -5 "$out/build_phase.log"; exit 1; }
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > "$out/bench_phase.json" 2> "$out/bench_phase.err" || { tail -5 "$out/bench_phase.err"; exit 1; }
grep "phase" "$out/bench_phase.err" | head -10
