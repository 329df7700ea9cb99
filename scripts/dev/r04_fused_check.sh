#!/bin/bash
# fused epoch kernel: parity tests that exercise it, bench line, kernel trace, in-kernel phase times
set -uo pipefail
tag=${1:-r04c}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_frontend.py -x -q -m gpu -k "virtual_shards or compact_records or handles_repeats or full_size or kkt or generators or stream or fit_" > "$out/pytest.log" 2>&1
rc=$?; tail -4 "$out/pytest.log"; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_fused.json" 2> "$out/bench_fused.err" || { tail -5 "$out/bench_fused.err"; exit 1; }
python3 - "$out/bench_fused.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fused", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("kernel_alone"), d.get("convergence"))
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 bench.py --no-cpu-baseline --no-convergence --steps 20 --warmup 5 > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
EXTRA_FLAGS=-DSGDNET_PHASE_TIMING ./build.sh > "$out/build_phase.log" 2>&1 || { tail -5 "$out/build_phase.log"; exit 1; }
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > "$out/bench_phase.json" 2> "$out/bench_phase.err" || { tail -5 "$out/bench_phase.err"; exit 1; }
grep "phase" "$out/bench_phase.err" | head -12
