#!/bin/bash
# first GPU run of the fused epoch kernel: parity tests that exercise it, then C4 A/B
set -uo pipefail
out=$PWD/gpurun_out/r04a
mkdir -p "$out"
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_shards or compact_records or handles_repeats or full_size or kkt" > "$out/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$out/pytest.log"
tail -5 "$out/pytest.log"
grep -q "failed\|error" "$out/pytest.log" && exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-epoch 0 > "$out/bench_unfused.json" 2> "$out/bench_unfused.err" || { tail -5 "$out/bench_unfused.err"; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-epoch 1 > "$out/bench_fused.json" 2> "$out/bench_fused.err" || { tail -5 "$out/bench_fused.err"; exit 1; }
python3 - <<'PY'
import json
for f in ("unfused", "fused"):
    d = json.loads(open(f"gpurun_out/r04a/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d.get("convergence"))
PY
