#!/bin/bash
# A/B of a compile-time variant on ONE box: builds the library with and without EXTRA flags and
# runs the same short measurement on each.  usage: scripts/ab_flags.sh "<flags>" <tag> [workload] [shards]
set -eo pipefail
FL="$1"; TAG="$2"; WL="${3:-C4}"; VS="${4:-1,8}"
mkdir -p gpurun_out/$TAG
EXTRA_FLAGS="$FL" ./build.sh > gpurun_out/$TAG/build_b.log 2>&1
timeout -k 10 400 python scripts/vshard_bench.py $WL $VS > gpurun_out/$TAG/variant.log 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_shards or batched" > gpurun_out/$TAG/variant_tests.log 2>&1 || true
./build.sh > gpurun_out/$TAG/build_a.log 2>&1
timeout -k 10 400 python scripts/vshard_bench.py $WL $VS > gpurun_out/$TAG/base.log 2>&1
echo "== variant ($FL)"; cat gpurun_out/$TAG/variant.log; tail -3 gpurun_out/$TAG/variant_tests.log; echo "== base"; cat gpurun_out/$TAG/base.log
